// Multi-GPU exchange of libjrender_hip.so: an RCCL communicator bound to a jr_ctx (one process per
// GPU).  The reference has no distributed code; what a sharded batch needs afterwards
// is (a) an all-gather of per-view image / gradient shards and (b) an all-reduce of the gradient of
// vertices shared by all views (demo2-deform.py:45 repeats ONE vertex set over the batch).  Both run
// device-to-device on the context's stream over xGMI — no host bounce, no PyTorch.
//
// librccl.so is dlopen'ed on first use so that single-GPU users never pay for loading it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

// The few RCCL / NCCL declarations this file needs, restated from the public API (nccl.h): the library is only
// dlopen'ed, so building libjrender_hip.so must not require the RCCL development headers either.
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;                     /* every other value is an error */
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
}

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/jrender_hip.h"

extern "C" void jr_set_error_(const char* msg);      // jr_api.cpp: thread-local message behind jr_last_error()

namespace {

int cfail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    jr_set_error_(buf);
    return 1;
}

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.handle) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return cfail("jr_comm: cannot load librccl.so (%s)", dlerror());
#define JR_SYM(field, name)                                                        \
    *(void**)(&g_rccl.field) = dlsym(h, name);                                     \
    if (!g_rccl.field) return cfail("jr_comm: librccl.so lacks %s", name)
    JR_SYM(GetUniqueId, "ncclGetUniqueId");
    JR_SYM(CommInitRank, "ncclCommInitRank");
    JR_SYM(CommDestroy, "ncclCommDestroy");
    JR_SYM(CommCount, "ncclCommCount");
    JR_SYM(CommUserRank, "ncclCommUserRank");
    JR_SYM(AllGather, "ncclAllGather");
    JR_SYM(AllReduce, "ncclAllReduce");
    JR_SYM(Broadcast, "ncclBroadcast");
    JR_SYM(GroupStart, "ncclGroupStart");
    JR_SYM(GroupEnd, "ncclGroupEnd");
    JR_SYM(GetErrorString, "ncclGetErrorString");
#undef JR_SYM
    g_rccl.handle = h;
    return 0;
}

#define JR_NCCL(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return cfail("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)
#define JR_HIPC(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return cfail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

}  // namespace

struct jr_comm {
    jr_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    double* scratch = nullptr;      // 2 doubles on the device: barrier token / scalar reductions
    double* h_scratch = nullptr;    // pinned mirror
};

static_assert(JR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

extern "C" {

int jr_comm_unique_id(void* id_host) {
    if (!id_host) return cfail("jr_comm_unique_id: NULL argument");
    if (load_rccl()) return 1;
    ncclUniqueId id;
    JR_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_host, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int jr_comm_create(jr_ctx* ctx, const void* id_host, int nranks, int rank, jr_comm** out) {
    if (!ctx || !id_host || !out) return cfail("jr_comm_create: NULL argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return cfail("jr_comm_create: rank %d of %d", rank, nranks);
    if (load_rccl()) return 1;
    JR_HIPC(hipSetDevice(jr_ctx_device(ctx)));
    jr_comm* c = new (std::nothrow) jr_comm();
    if (!c) return cfail("out of host memory");
    c->ctx = ctx; c->nranks = nranks; c->rank = rank;
    ncclUniqueId id;
    memcpy(id.internal, id_host, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return cfail("ncclCommInitRank(rank %d of %d, GPU %d) failed: %s", rank, nranks, jr_ctx_device(ctx),
                     g_rccl.GetErrorString(r));
    }
    // What RCCL itself says about the communicator must be what the caller asked for: jr_comm_size / jr_comm_rank return
    // THESE values (bench.py prints the size as `rccl_ranks`, so that a scaling record shows RCCL really saw N ranks).
    int n_seen = -1, r_seen = -1;
    if (g_rccl.CommCount(c->comm, &n_seen) != ncclSuccess || g_rccl.CommUserRank(c->comm, &r_seen) != ncclSuccess ||
        n_seen != nranks || r_seen != rank) {
        (void)g_rccl.CommDestroy(c->comm);
        delete c;
        return cfail("jr_comm_create: RCCL reports rank %d of %d for the communicator created as rank %d of %d", r_seen, n_seen, rank, nranks);
    }
    c->nranks = n_seen; c->rank = r_seen;
    // a communicator that cannot get its scratch words is torn down again: the peers must not be left holding a
    // half-created one, and neither the struct nor the live ncclComm may leak
    hipError_t he = hipMalloc((void**)&c->scratch, sizeof(double) * 2);
    if (he == hipSuccess) he = hipHostMalloc((void**)&c->h_scratch, sizeof(double) * 2, hipHostMallocDefault);
    if (he != hipSuccess) {
        (void)g_rccl.CommDestroy(c->comm);
        if (c->scratch) (void)hipFree(c->scratch);
        delete c;
        return cfail("jr_comm_create: scratch allocation failed: %s", hipGetErrorString(he));
    }
    *out = c;
    return 0;
}

int jr_comm_destroy(jr_comm* c) {
    if (!c) return 0;
    (void)hipSetDevice(jr_ctx_device(c->ctx));
    (void)hipStreamSynchronize((hipStream_t)jr_ctx_stream(c->ctx));
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    (void)hipFree(c->scratch);
    (void)hipHostFree(c->h_scratch);
    delete c;
    return 0;
}

// ncclCommUserRank / ncclCommCount of the live communicator (queried again: not the values stored at creation)
int jr_comm_rank(const jr_comm* c) {
    int r = -1;
    if (!c || !c->comm || g_rccl.CommUserRank(c->comm, &r) != ncclSuccess) return -1;
    return r;
}
int jr_comm_size(const jr_comm* c) {
    int n = 0;
    if (!c || !c->comm || g_rccl.CommCount(c->comm, &n) != ncclSuccess) return 0;
    return n;
}

int jr_comm_all_gather(jr_comm* c, const void* send, void* recv, size_t bytes_per_rank) {
    if (!c) return cfail("jr_comm_all_gather: NULL communicator");
    if (bytes_per_rank == 0) return 0;                    // an empty shard has no buffers to speak of
    if (!send || !recv) return cfail("jr_comm_all_gather: NULL buffer");
    JR_HIPC(hipSetDevice(jr_ctx_device(c->ctx)));
    hipStream_t st = (hipStream_t)jr_ctx_stream(c->ctx);
    // whole floats where possible (RCCL's copy kernels move wider elements faster than bytes)
    if (bytes_per_rank % 4 == 0)
        JR_NCCL(g_rccl.AllGather(send, recv, bytes_per_rank / 4, ncclFloat32, c->comm, st));
    else
        JR_NCCL(g_rccl.AllGather(send, recv, bytes_per_rank, ncclUint8, c->comm, st));
    return 0;
}

int jr_comm_all_gather_v(jr_comm* c, const void* send, void* recv, const size_t* bytes_of_rank) {
    if (!c || !recv || !bytes_of_rank) return cfail("jr_comm_all_gather_v: NULL argument");
    JR_HIPC(hipSetDevice(jr_ctx_device(c->ctx)));
    hipStream_t st = (hipStream_t)jr_ctx_stream(c->ctx);
    // uneven shards: one broadcast per rank, fused in a group (no padding copies)
    JR_NCCL(g_rccl.GroupStart());
    size_t off = 0;
    for (int r = 0; r < c->nranks; r++) {
        const size_t n = bytes_of_rank[r];
        if (n) {
            char* dst = (char*)recv + off;
            const void* src = r == c->rank ? send : (const void*)dst;
            if (r == c->rank && !send) { (void)g_rccl.GroupEnd(); return cfail("jr_comm_all_gather_v: NULL send buffer"); }
            ncclResult_t rc = g_rccl.Broadcast(src, dst, n, ncclUint8, r, c->comm, st);
            if (rc != ncclSuccess) { (void)g_rccl.GroupEnd(); return cfail("ncclBroadcast failed: %s", g_rccl.GetErrorString(rc)); }
        }
        off += n;
    }
    JR_NCCL(g_rccl.GroupEnd());
    return 0;
}

int jr_comm_all_reduce_f32(jr_comm* c, const float* send, float* recv, size_t count, int op) {
    if (!c || !send || !recv) return cfail("jr_comm_all_reduce_f32: NULL argument");
    if (op != JR_REDUCE_SUM && op != JR_REDUCE_MAX) return cfail("jr_comm_all_reduce_f32: op must be JR_REDUCE_SUM or JR_REDUCE_MAX");
    JR_HIPC(hipSetDevice(jr_ctx_device(c->ctx)));
    if (count == 0) return 0;
    JR_NCCL(g_rccl.AllReduce(send, recv, count, ncclFloat32, op == JR_REDUCE_SUM ? ncclSum : ncclMax, c->comm,
                             (hipStream_t)jr_ctx_stream(c->ctx)));
    return 0;
}

int jr_comm_all_reduce_host_f64(jr_comm* c, double* values_host, int count, int op) {
    if (!c || !values_host) return cfail("jr_comm_all_reduce_host_f64: NULL argument");
    if (count < 1 || count > 2) return cfail("jr_comm_all_reduce_host_f64: count must be 1 or 2");
    if (op != JR_REDUCE_SUM && op != JR_REDUCE_MAX) return cfail("jr_comm_all_reduce_host_f64: bad op");
    JR_HIPC(hipSetDevice(jr_ctx_device(c->ctx)));
    hipStream_t st = (hipStream_t)jr_ctx_stream(c->ctx);
    memcpy(c->h_scratch, values_host, sizeof(double) * count);
    JR_HIPC(hipMemcpyAsync(c->scratch, c->h_scratch, sizeof(double) * count, hipMemcpyHostToDevice, st));
    JR_NCCL(g_rccl.AllReduce(c->scratch, c->scratch, (size_t)count, ncclFloat64, op == JR_REDUCE_SUM ? ncclSum : ncclMax,
                             c->comm, st));
    JR_HIPC(hipMemcpyAsync(c->h_scratch, c->scratch, sizeof(double) * count, hipMemcpyDeviceToHost, st));
    JR_HIPC(hipStreamSynchronize(st));
    memcpy(values_host, c->h_scratch, sizeof(double) * count);
    return 0;
}

int jr_comm_barrier(jr_comm* c) {
    double token = 1.0;
    return jr_comm_all_reduce_host_f64(c, &token, 1, JR_REDUCE_SUM);
}

}  // extern "C"
