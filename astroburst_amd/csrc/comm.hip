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
#include "ab_common.hpp"

#include <dlfcn.h>

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

}  // namespace

struct ab_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1, device = 0;
    uint64_t collectives = 0;  // issued so far (bench / tests report it)
};

#define AB_NCCL(ctx, r, call)                                                                                            \
    do {                                                                                                                 \
        int e_ = (call);                                                                                                 \
        if (e_ != ncclSuccess) return ab_set_error((ctx), AB_ERR_COMM, "%s failed: %s", #call, (r)->GetErrorString(e_)); \
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
        out_comms[i] = c;
    }
    return AB_OK;
} AB_CATCH_NOCTX

void ab_comm_destroy(ab_comm *c) {
    if (!c) return;
    const Rccl *r = rccl();
    if (r && c->comm) {
        (void)hipSetDevice(c->device);
        (void)r->CommDestroy(c->comm);
    }
    delete c;
}

int ab_comm_rank(const ab_comm *c) { return c ? c->rank : 0; }
int ab_comm_size(const ab_comm *c) { return c ? c->size : 1; }
uint64_t ab_comm_collectives_issued(const ab_comm *c) { return c ? c->collectives : 0; }

int ab_comm_group_start(void) try {
    const Rccl *r = rccl();
    return (r && r->GroupStart() == ncclSuccess) ? AB_OK : AB_ERR_COMM;
} AB_CATCH_NOCTX
int ab_comm_group_end(void) try {
    const Rccl *r = rccl();
    return (r && r->GroupEnd() == ncclSuccess) ? AB_OK : AB_ERR_COMM;
} AB_CATCH_NOCTX

// in place, on the context's stream, asynchronous; a NULL communicator is a world of one (no-op)
int ab_comm_allreduce(ab_ctx *ctx, ab_comm *c, void *buf_dev, size_t count, int dtype, int op) try {
    if (!ctx) return AB_ERR_INVALID;
    if (!c) return AB_OK;
    AB_CHECK(ctx, buf_dev && count > 0, "null or empty all-reduce buffer");
    AB_CHECK(ctx, c->device == ctx->device, "communicator is bound to device %d, context to %d", c->device, ctx->device);
    size_t elem = 0;
    const int nt = to_nccl_type(dtype, &elem);
    AB_CHECK(ctx, nt >= 0, "bad ab_dtype %d", dtype);
    const int nop = op == AB_RED_SUM ? ncclSum : (op == AB_RED_MAX ? ncclMax : (op == AB_RED_MIN ? ncclMin : -1));
    AB_CHECK(ctx, nop >= 0, "bad ab_redop %d", op);
    const Rccl *r = rccl();
    if (!r) return ab_set_error(ctx, AB_ERR_COMM, "%s", g_rccl.why.c_str());
    AB_HIP(ctx, hipSetDevice(ctx->device));
    AB_NCCL(ctx, r, r->AllReduce(buf_dev, buf_dev, count, nt, nop, c->comm, ctx->stream));
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
    AB_CHECK(ctx, c->device == ctx->device, "communicator is bound to device %d, context to %d", c->device, ctx->device);
    const Rccl *r = rccl();
    if (!r) return ab_set_error(ctx, AB_ERR_COMM, "%s", g_rccl.why.c_str());
    AB_HIP(ctx, hipSetDevice(ctx->device));
    AB_NCCL(ctx, r, r->AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, c->comm, ctx->stream));
    c->collectives++;
    return AB_OK;
} AB_CATCH(ctx)

int ab_comm_broadcast(ab_ctx *ctx, ab_comm *c, void *buf_dev, size_t bytes, int root) try {
    if (!ctx) return AB_ERR_INVALID;
    if (!c) return AB_OK;
    AB_CHECK(ctx, buf_dev && bytes > 0 && root >= 0 && root < c->size, "bad broadcast arguments");
    const Rccl *r = rccl();
    if (!r) return ab_set_error(ctx, AB_ERR_COMM, "%s", g_rccl.why.c_str());
    AB_HIP(ctx, hipSetDevice(ctx->device));
    AB_NCCL(ctx, r, r->Broadcast(buf_dev, buf_dev, bytes, ncclUint8, root, c->comm, ctx->stream));
    c->collectives++;
    return AB_OK;
} AB_CATCH(ctx)

}  // extern "C"
