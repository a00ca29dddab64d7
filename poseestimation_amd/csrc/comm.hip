// Multi-GPU collect of the skeleton buffers without any framework in the process: a thin
// communicator over RCCL (loaded at run time, so that the library itself carries no link
// dependency on it) and pd_gather -- an all-gather of the per-rank byte counts followed by
// grouped sends / receives of the packed buffers to the root rank, which copies them to the
// caller's host array in rank-major order (= image order of a contiguously sharded batch).
#include <dlfcn.h>
#include <stdlib.h>

#include <vector>

#include "common.h"

// The few RCCL declarations used (rccl.h: ncclResult_t, ncclUniqueId, ncclDataType_t).
typedef int rc_result;
typedef struct {
    char internal[128];
} rc_unique_id;
enum { RC_UINT8 = 1, RC_INT64 = 4 };
typedef void* rc_comm;

struct RcclApi {
    void* lib;
    rc_result (*GetUniqueId)(rc_unique_id*);
    rc_result (*CommInitRank)(rc_comm*, int, rc_unique_id, int);
    rc_result (*CommDestroy)(rc_comm);
    rc_result (*AllGather)(const void*, void*, size_t, int, rc_comm, hipStream_t);
    rc_result (*Send)(const void*, size_t, int, int, rc_comm, hipStream_t);
    rc_result (*Recv)(void*, size_t, int, int, rc_comm, hipStream_t);
    rc_result (*GroupStart)(void);
    rc_result (*GroupEnd)(void);
    const char* (*GetErrorString)(rc_result);
};

static RcclApi g_rccl;
static std::mutex g_rccl_mu;

// PD_RCCL_LIBRARY names the library file; otherwise the one already in the process (a
// framework may have brought its own) or the system's librccl.so.1 is taken.
static int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.lib) return PD_OK;
    const char* names[] = {getenv("PD_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) {
        if (!n || !*n) continue;
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) {
        pd_set_global_error("pd_comm: librccl.so not found (%s)", dlerror());
        return PD_E_STATE;
    }
    RcclApi a;
    a.lib = lib;
#define SYM(field, name)                                          \
    *(void**)(&a.field) = dlsym(lib, name);                       \
    if (!a.field) {                                               \
        pd_set_global_error("pd_comm: %s missing in librccl", name); \
        dlclose(lib);                                             \
        return PD_E_STATE;                                        \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl = a;
    return PD_OK;
}

struct pd_comm {
    int world, rank, device;
    rc_comm comm;
    hipStream_t stream;
    int64_t* counts_dev;   // [world][slots] bytes of every rank's buffers (slots = PD_GATHER_MAX_CTX)
    int64_t* counts_host;  // pinned mirror
    unsigned char* recv_dev;  // root: the gathered buffers, back to back
    size_t recv_cap;
    char err[512];
};

#define PD_GATHER_MAX_CTX 16

#define CM_FAIL(c, code, ...)                              \
    do {                                                   \
        snprintf((c)->err, sizeof((c)->err), __VA_ARGS__); \
        pd_set_global_error("%s", (c)->err);               \
        return (code);                                     \
    } while (0)
#define CM_HIP(c, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) CM_FAIL(c, PD_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define CM_RCCL(c, call)                                                                             \
    do {                                                                                             \
        rc_result r_ = (call);                                                                       \
        if (r_ != 0) CM_FAIL(c, PD_E_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(r_));      \
    } while (0)

extern "C" int pd_comm_unique_id(void* id128) {
    if (!id128) return PD_E_INVALID;
    int rc = load_rccl();
    if (rc != PD_OK) return rc;
    rc_unique_id id;
    const rc_result r = g_rccl.GetUniqueId(&id);
    if (r != 0) {
        pd_set_global_error("ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
        return PD_E_HIP;
    }
    memcpy(id128, id.internal, sizeof(id.internal));
    return PD_OK;
}

extern "C" int pd_comm_destroy(pd_comm* c) {
    if (!c) return PD_E_INVALID;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->counts_dev) (void)hipFree(c->counts_dev);
    if (c->counts_host) (void)hipHostFree(c->counts_host);
    if (c->recv_dev) (void)hipFree(c->recv_dev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PD_OK;
}

extern "C" int pd_comm_create(const void* id128, int world, int rank, int device_id, pd_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return PD_E_INVALID;
    *out = nullptr;
    int rc = load_rccl();
    if (rc != PD_OK) return rc;
    pd_comm* c = new (std::nothrow) pd_comm;
    if (!c) return PD_E_HIP;
    memset(c, 0, sizeof(*c));
    c->world = world;
    c->rank = rank;
    c->device = device_id;
#define CC(call)                         \
    do {                                 \
        int rc_ = [&]() -> int {         \
            call;                        \
            return PD_OK;                \
        }();                             \
        if (rc_ != PD_OK) {              \
            pd_comm_destroy(c);          \
            return rc_;                  \
        }                                \
    } while (0)
    CC(CM_HIP(c, hipSetDevice(device_id)));
    CC(CM_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)));
    const size_t nb = (size_t)world * PD_GATHER_MAX_CTX * sizeof(int64_t);
    CC(CM_HIP(c, hipMalloc((void**)&c->counts_dev, nb)));
    CC(CM_HIP(c, hipHostMalloc((void**)&c->counts_host, nb, hipHostMallocDefault)));
    rc_unique_id id;
    memcpy(id.internal, id128, sizeof(id.internal));
    CC(CM_RCCL(c, g_rccl.CommInitRank(&c->comm, world, id, rank)));
#undef CC
    *out = c;
    return PD_OK;
}

extern "C" const char* pd_comm_last_error(pd_comm* c) { return c ? c->err : pd_last_error(nullptr); }

// Bytes of a packed skeleton buffer, from its header (B, n_peaks, n_conns, n_persons).
static size_t blob_bytes(const int64_t* head) {
    return pd_res_layout((int)head[0]).off_payload + (size_t)head[1] * sizeof(pd_peak) + (size_t)head[2] * sizeof(pd_conn) +
           (size_t)head[3] * 20 * sizeof(double);
}

extern "C" int pd_gather(pd_comm* c, pd_ctx* const* ctxs, int n_ctx, int root, void* recv_host, uint64_t recv_capacity,
                         uint64_t* sizes, uint64_t* total_bytes) {
    if (!c) return PD_E_INVALID;
    if (!ctxs || n_ctx < 1 || n_ctx > PD_GATHER_MAX_CTX || root < 0 || root >= c->world)
        CM_FAIL(c, PD_E_INVALID, "pd_gather: 1..%d contexts per rank, root within the communicator", PD_GATHER_MAX_CTX);
    CM_HIP(c, hipSetDevice(c->device));
    // every context's decode must have reached its skeleton buffer; its header gives the size
    int64_t mine[PD_GATHER_MAX_CTX];
    memset(mine, 0, sizeof(mine));
    int short_ctx = -1, bad_ctx = -1;
    long long short_need = 0;
    for (int i = 0; i < n_ctx; ++i) {
        pd_ctx* x = ctxs[i];
        if (!x || !x->skel_dst) CM_FAIL(c, PD_E_STATE, "pd_gather: context %d has no skeleton buffer (pd_set_skeleton_buffer)", i);
        if (x->cfg.device_id != c->device) CM_FAIL(c, PD_E_INVALID, "pd_gather: context %d is on another device", i);
        if (x->worker && pd_flush(x) != PD_OK)  // decodes queued with pd_submit have been launched
            CM_FAIL(c, PD_E_STATE, "pd_gather: context %d: %s", i, x->err);
        CM_HIP(c, hipStreamSynchronize(x->stream));
        int64_t head[8];
        CM_HIP(c, hipMemcpy(head, x->skel_dst, sizeof(head), hipMemcpyDeviceToHost));
        const bool sane = head[0] >= 1 && head[0] <= x->cfg.max_batch && head[1] >= 0 && head[2] >= 0 && head[3] >= 0;
        if (sane && head[5] != 0) {
            // the buffer was too small for the payload: the header (which says so) and the count
            // tables still travel, so that no rank is left waiting; reported after the exchange
            short_ctx = i;
            short_need = (long long)head[5];
            mine[i] = (int64_t)pd_res_layout((int)head[0]).off_payload;
        } else if (sane) {
            mine[i] = (int64_t)blob_bytes(head);
        }
        if (!sane || (uint64_t)mine[i] > (uint64_t)x->skel_cap) {
            // no decode has written this buffer (or not this one): a send sized from its header would
            // read past its end.  The size -1 travels in the count exchange, so that every rank
            // gives up together instead of waiting for this one.
            bad_ctx = i;
            mine[i] = -1;
        }
    }
    // 1. byte counts of every rank
    CM_HIP(c, hipMemcpyAsync(c->counts_dev + (size_t)c->rank * PD_GATHER_MAX_CTX, mine, sizeof(mine), hipMemcpyHostToDevice,
                             c->stream));
    CM_RCCL(c, g_rccl.AllGather(c->counts_dev + (size_t)c->rank * PD_GATHER_MAX_CTX, c->counts_dev, PD_GATHER_MAX_CTX, RC_INT64,
                                c->comm, c->stream));
    CM_HIP(c, hipMemcpyAsync(c->counts_host, c->counts_dev, (size_t)c->world * PD_GATHER_MAX_CTX * sizeof(int64_t),
                             hipMemcpyDeviceToHost, c->stream));
    CM_HIP(c, hipStreamSynchronize(c->stream));
    if (bad_ctx >= 0)
        CM_FAIL(c, PD_E_STATE, "pd_gather: context %d: its skeleton buffer (%llu bytes) does not hold the results of a decode",
                bad_ctx, (unsigned long long)ctxs[bad_ctx]->skel_cap);
    uint64_t total = 0;
    for (int r = 0; r < c->world; ++r)
        for (int i = 0; i < PD_GATHER_MAX_CTX; ++i) {
            const int64_t v = c->counts_host[(size_t)r * PD_GATHER_MAX_CTX + i];
            if (v < 0) CM_FAIL(c, PD_E_STATE, "pd_gather: rank %d has a skeleton buffer that holds no decode results", r);
            if ((v != 0) != (i < n_ctx))
                CM_FAIL(c, PD_E_INVALID, "pd_gather: rank %d brought another number of contexts than this rank (%d)", r, n_ctx);
            total += (uint64_t)((v + 15) & ~(int64_t)15);   // every buffer starts 16-byte aligned
        }
    if (total_bytes) *total_bytes = total;
    const bool is_root = c->rank == root;
    // the root's receive area; whether it could be had travels to every rank before anyone sends
    int64_t ready = 1;
    if (is_root && c->recv_cap < total) {
        if (c->recv_dev) (void)hipFree(c->recv_dev);
        c->recv_dev = nullptr;
        c->recv_cap = 0;
        if (hipMalloc((void**)&c->recv_dev, total) == hipSuccess)
            c->recv_cap = total;
        else
            ready = 0;
    }
    if (c->world > 1) {
        int64_t* flags = c->counts_dev;  // (the counts have been copied out)
        CM_HIP(c, hipMemcpyAsync(flags + c->rank, &ready, sizeof(ready), hipMemcpyHostToDevice, c->stream));
        CM_RCCL(c, g_rccl.AllGather(flags + c->rank, flags, 1, RC_INT64, c->comm, c->stream));
        int64_t root_ready = 0;
        CM_HIP(c, hipMemcpyAsync(&root_ready, flags + root, sizeof(root_ready), hipMemcpyDeviceToHost, c->stream));
        CM_HIP(c, hipStreamSynchronize(c->stream));
        ready = root_ready;
    }
    if (!ready) CM_FAIL(c, PD_E_HIP, "pd_gather: rank %d could not allocate %llu bytes for the gathered buffers", root, (unsigned long long)total);
    // 2. the buffers themselves: one send per context, one receive per (rank, context) on the root.
    // A group that was opened is always closed; the first failure inside it is reported afterwards.
    rc_result first = g_rccl.GroupStart();
    if (first != 0) CM_FAIL(c, PD_E_HIP, "ncclGroupStart failed: %s", g_rccl.GetErrorString(first));
    for (int i = 0; i < n_ctx && first == 0; ++i)
        first = g_rccl.Send(ctxs[i]->skel_dst, (size_t)mine[i], RC_UINT8, root, c->comm, c->stream);
    if (is_root) {
        size_t off = 0;
        for (int r = 0; r < c->world; ++r)
            for (int i = 0; i < n_ctx; ++i) {
                const size_t v = (size_t)c->counts_host[(size_t)r * PD_GATHER_MAX_CTX + i];
                if (first == 0) first = g_rccl.Recv(c->recv_dev + off, v, RC_UINT8, r, c->comm, c->stream);
                if (sizes) sizes[(size_t)r * n_ctx + i] = v;
                off += (v + 15) & ~(size_t)15;
            }
    }
    const rc_result ended = g_rccl.GroupEnd();
    if (first == 0) first = ended;
    if (first != 0) CM_FAIL(c, PD_E_HIP, "pd_gather: the exchange failed: %s", g_rccl.GetErrorString(first));
    const bool fits = !is_root || (recv_host != nullptr && recv_capacity >= total);
    if (is_root && fits) CM_HIP(c, hipMemcpyAsync(recv_host, c->recv_dev, total, hipMemcpyDeviceToHost, c->stream));
    CM_HIP(c, hipStreamSynchronize(c->stream));
    if (short_ctx >= 0)
        CM_FAIL(c, PD_E_OVERFLOW, "pd_gather: context %d: its skeleton buffer holds %llu bytes, the results need %lld "
                                  "(header and count tables were sent)",
                short_ctx, (unsigned long long)ctxs[short_ctx]->skel_cap, short_need);
    if (!fits)
        CM_FAIL(c, PD_E_OVERFLOW, "pd_gather: the gathered buffers take %llu bytes, room for %llu (the exchange itself "
                                  "has completed on every rank)",
                (unsigned long long)total, (unsigned long long)recv_capacity);
    return PD_OK;
}

// ---------------------------------------------------------------------------------
// The root's way to its host: of every gathered skeleton buffer only the bytes in use.  The slices of
// a round are fixed-size (their capacity is what the largest result may take; a typical one fills half
// of it), and at eight ranks a plain copy of all of them is most of what a PCIe link carries.  One
// workgroup per slice reads the slice's header, works out where its payload ends and stores that much
// to the same offset of the destination, which may be host memory the device can write (pinned).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_copy_used(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, unsigned long long capacity) {
    const unsigned char* s = src + (size_t)blockIdx.x * capacity;
    unsigned char* d = dst + (size_t)blockIdx.x * capacity;
    const long long* head = (const long long*)s;
    const long long B = head[0], n_peaks = head[1], n_conns = head[2], n_persons = head[3], need = head[5];
    unsigned long long used = 64;   // a slice no decode has written: its header alone (the reader rejects it)
    if (B >= 1 && B <= (1 << 20) && n_peaks >= 0 && n_conns >= 0 && n_persons >= 0) {
        const unsigned long long tables = pd_res_layout((int)B).off_payload;
        used = need != 0 ? tables
                         : tables + (unsigned long long)n_peaks * sizeof(pd_peak) + (unsigned long long)n_conns * sizeof(pd_conn) +
                               (unsigned long long)n_persons * 20 * sizeof(double);
    }
    if (used > capacity) used = capacity;
    const size_t n16 = (size_t)((used + 15) / 16);  // (the capacity is a multiple of 16: rounding up stays inside the slice)
    const uint4* s4 = (const uint4*)s;
    uint4* d4 = (uint4*)d;
    for (size_t i = threadIdx.x; i < n16; i += 256) d4[i] = s4[i];
}

extern "C" int pd_copy_used_skeletons(const void* src_dev, void* dst, int n_slices, uint64_t capacity, void* stream) {
    if (!src_dev || !dst || n_slices < 1 || capacity < 64 || (capacity & 15) || ((uintptr_t)src_dev & 15) || ((uintptr_t)dst & 15)) {
        pd_set_global_error("pd_copy_used_skeletons: %d slices of %llu bytes (a multiple of 16, at least 64) at 16-byte aligned addresses",
                            n_slices, (unsigned long long)capacity);
        return PD_E_INVALID;
    }
    hipLaunchKernelGGL(k_copy_used, dim3((unsigned)n_slices), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src_dev,
                       (unsigned char*)dst, (unsigned long long)capacity);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pd_set_global_error("pd_copy_used_skeletons: launch failed: %s", hipGetErrorString(e));
        return PD_E_HIP;
    }
    return PD_OK;
}
