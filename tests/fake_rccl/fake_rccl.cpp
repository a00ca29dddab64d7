// Test double for the RCCL entry points pd_gather uses, so that its protocol -- the count
// exchange, the grouped sends of every rank and the root's receives from OTHER ranks -- can run
// with a world of several ranks inside one process on one GPU: every rank is a host thread, a
// "communicator" is a slot in a world that all threads reach through the unique id, and data moves
// with device-to-device copies.  Collectives meet at a barrier, as the real ones do.  TEST
// INFRASTRUCTURE ONLY: the library loads it when PD_RCCL_LIBRARY names it; the product never does.
#include <hip/hip_runtime.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Msg {
    const void* ptr;
    size_t bytes;
};

struct World {
    int n = 0, arrived = 0, generation = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<const void*> slot;                      // all-gather: every rank's send buffer
    std::map<std::pair<int, int>, std::deque<Msg>> box;  // (src, dst) -> posted sends, in order
    void barrier() {
        std::unique_lock<std::mutex> lock(mu);
        const int gen = generation;
        if (++arrived == n) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return generation != gen; });
        }
    }
};

struct Comm {
    World* world;
    int rank;
};

struct Op {
    bool send;
    void* ptr;
    size_t bytes;
    int peer;
    Comm* comm;
    hipStream_t stream;
};

std::mutex g_mu;
std::map<std::string, World*> g_worlds;
int g_next_id = 1;
thread_local int t_group = 0;
thread_local std::vector<Op> t_ops;

size_t type_size(int dtype) { return dtype == 1 ? 1 : (dtype == 4 ? 8 : 0); }  // ncclUint8, ncclInt64

int run_ops() {
    if (t_ops.empty()) return 0;
    Comm* c = t_ops[0].comm;
    World* w = c->world;
    for (const Op& op : t_ops)
        if (hipStreamSynchronize(op.stream) != hipSuccess) return 1;  // what is sent has been written
    {
        std::lock_guard<std::mutex> lock(w->mu);
        for (const Op& op : t_ops)
            if (op.send) w->box[{c->rank, op.peer}].push_back(Msg{op.ptr, op.bytes});
    }
    w->barrier();  // every rank has posted its sends
    int rc = 0;
    for (const Op& op : t_ops) {
        if (op.send) continue;
        Msg m{nullptr, 0};
        {
            std::lock_guard<std::mutex> lock(w->mu);
            auto& q = w->box[{op.peer, c->rank}];
            if (q.empty()) {
                rc = 3;  // a receive without a matching send
                continue;
            }
            m = q.front();
            q.pop_front();
        }
        if (m.bytes != op.bytes) {
            rc = 4;  // sizes of a send and its receive differ
            continue;
        }
        if (m.bytes && hipMemcpy(op.ptr, m.ptr, m.bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = 1;
    }
    w->barrier();  // nobody reuses a send buffer before it has been read
    t_ops.clear();
    return rc;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(void* id) {
    std::lock_guard<std::mutex> lock(g_mu);
    memset(id, 0, 128);
    snprintf((char*)id, 128, "fake-world-%d", g_next_id++);
    return 0;
}

struct IdByValue {
    char internal[128];
};

int ncclCommInitRank(void** comm, int nranks, IdByValue id, int rank) {
    std::lock_guard<std::mutex> lock(g_mu);
    World*& w = g_worlds[std::string(id.internal)];
    if (!w) {
        w = new World;
        w->n = nranks;
        w->slot.assign(nranks, nullptr);
    }
    if (w->n != nranks || rank < 0 || rank >= nranks) return 2;
    *comm = new Comm{w, rank};
    return 0;
}

int ncclCommDestroy(void* comm) {
    delete (Comm*)comm;
    return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    World* w = c->world;
    const size_t bytes = count * type_size(dtype);
    if (bytes == 0) return 2;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    {
        std::lock_guard<std::mutex> lock(w->mu);
        w->slot[c->rank] = send;
    }
    w->barrier();
    int rc = 0;
    for (int r = 0; r < w->n; ++r) {
        void* dst = (char*)recv + (size_t)r * bytes;
        if (dst != w->slot[r] && hipMemcpy(dst, w->slot[r], bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = 1;
    }
    w->barrier();
    return rc;
}

int ncclGroupStart(void) {
    ++t_group;
    return 0;
}

int ncclGroupEnd(void) {
    if (--t_group > 0) return 0;
    return run_ops();
}

int ncclSend(const void* ptr, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
    t_ops.push_back(Op{true, (void*)ptr, count * type_size(dtype), peer, (Comm*)comm, stream});
    return t_group > 0 ? 0 : run_ops();
}

int ncclRecv(void* ptr, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
    t_ops.push_back(Op{false, ptr, count * type_size(dtype), peer, (Comm*)comm, stream});
    return t_group > 0 ? 0 : run_ops();
}

const char* ncclGetErrorString(int rc) {
    switch (rc) {
        case 0: return "ok";
        case 1: return "fake rccl: a HIP call failed";
        case 2: return "fake rccl: bad argument";
        case 3: return "fake rccl: receive without a matching send";
        case 4: return "fake rccl: send and receive sizes differ";
    }
    return "fake rccl: error";
}
}
