// loopback_ccl.cpp — TEST DOUBLE for the ten RCCL entry points csrc/comm_api.inc binds (BSG_RCCL_LIBRARY points at the
// built .so).  Ranks are THREADS of one process (or several devices-entries of one context in one thread), every buffer is
// in the same address space, and an exchange is plain device-to-device copies behind host-side rendezvous.  It exists so
// that ONE GPU can run the world > 1 slice schedule of bsg_or_allreduce (send slice j to rank j / OR / all-gather in
// place), whose index arithmetic is otherwise only executed on a multi-GPU node.  Not a model of RCCL's performance or of
// its stream semantics: every operation completes before the call (or ncclGroupEnd) returns.
//
//   hipcc -O1 -std=c++17 -shared -fPIC -o tests/_build/libloopback_ccl.so tests/loopback_ccl.cpp
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct World {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0;
    // point to point: (src, dst) -> what src offers; erased by dst once copied
    struct Offer { const void *ptr; size_t bytes; };
    std::map<std::pair<int, int>, Offer> box;
    // all-gather: what every rank offers in the current round, and how many have finished copying
    std::vector<const void *> ag;
    size_t ag_bytes = 0;
    int ag_posted = 0, ag_done = 0;
    uint64_t ag_round = 0;
};

struct Comm { int rank; std::shared_ptr<World> w; };

std::mutex g_mu;
std::map<uint64_t, std::shared_ptr<World>> g_worlds;
uint64_t g_next = 1;

struct Op { int kind; const void *src; void *dst; size_t bytes; int peer; Comm *c; hipStream_t s; };   // 0 send, 1 recv, 2 all-gather
thread_local int tl_depth = 0;
thread_local std::vector<Op> tl_ops;

ncclResult_t copy_now(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (bytes == 0 || dst == src) return ncclSuccess;
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    return hipStreamSynchronize(s) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t run(std::vector<Op> &ops)
{
    // what the senders offer must be complete: drain every stream named first
    for (const Op &o : ops) if (hipStreamSynchronize(o.s) != hipSuccess) return ncclUnhandledCudaError;
    // 1. post every send and every all-gather contribution
    std::vector<uint64_t> round(ops.size(), 0);
    for (size_t i = 0; i < ops.size(); ++i) {
        const Op &o = ops[i];
        World &w = *o.c->w;
        std::unique_lock<std::mutex> lk(w.mu);
        if (o.kind == 0) {
            w.cv.wait(lk, [&] { return w.box.find({o.c->rank, o.peer}) == w.box.end(); });     // (a previous offer still unread)
            w.box[{o.c->rank, o.peer}] = World::Offer{o.src, o.bytes};
        } else if (o.kind == 2) {
            w.cv.wait(lk, [&] { return w.ag_posted < w.world && w.ag[o.c->rank] == nullptr; });   // (the previous round has been cleared)
            w.ag[o.c->rank] = o.src;
            w.ag_bytes = o.bytes;
            round[i] = w.ag_round;
            ++w.ag_posted;
        }
        w.cv.notify_all();
    }
    // 2. receive
    for (const Op &o : ops) {
        World &w = *o.c->w;
        if (o.kind == 1) {
            World::Offer of{};
            {
                std::unique_lock<std::mutex> lk(w.mu);
                w.cv.wait(lk, [&] { return w.box.find({o.peer, o.c->rank}) != w.box.end(); });
                of = w.box[{o.peer, o.c->rank}];
            }
            if (of.bytes != o.bytes) return ncclInvalidArgument;
            // LOOPBACK_CCL_BREAK=1 drops what the last rank sends (the suite's check that a wrong exchange is noticed)
            static const bool broken = getenv("LOOPBACK_CCL_BREAK") != nullptr;
            if (!(broken && o.peer == w.world - 1))
                if (ncclResult_t r = copy_now(o.dst, of.ptr, o.bytes, o.s)) return r;
            std::unique_lock<std::mutex> lk(w.mu);
            w.box.erase({o.peer, o.c->rank});
            w.cv.notify_all();
        } else if (o.kind == 2) {
            std::vector<const void *> from;
            {
                std::unique_lock<std::mutex> lk(w.mu);
                w.cv.wait(lk, [&] { return w.ag_posted == w.world; });
                from = w.ag;
            }
            for (int j = 0; j < w.world; ++j)
                if (ncclResult_t r = copy_now(static_cast<char *>(o.dst) + (size_t)j * o.bytes, from[j], o.bytes, o.s)) return r;
            std::unique_lock<std::mutex> lk(w.mu);
            if (++w.ag_done == w.world) {                    // the last reader opens the next round
                std::fill(w.ag.begin(), w.ag.end(), nullptr);
                w.ag_posted = w.ag_done = 0;
                ++w.ag_round;
            }
            w.cv.notify_all();
        }
    }
    // 3. a send returns when it has been read, an all-gather when every rank has read every contribution: the caller may
    //    reuse or free its buffers right after
    for (size_t i = 0; i < ops.size(); ++i) {
        const Op &o = ops[i];
        World &w = *o.c->w;
        std::unique_lock<std::mutex> lk(w.mu);
        if (o.kind == 0) w.cv.wait(lk, [&] { return w.box.find({o.c->rank, o.peer}) == w.box.end(); });
        else if (o.kind == 2) w.cv.wait(lk, [&] { return w.ag_round > round[i]; });
    }
    return ncclSuccess;
}

ncclResult_t submit(const Op &o)
{
    tl_ops.push_back(o);
    if (tl_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(tl_ops);
    return run(ops);
}

size_t width(ncclDataType_t t) { return t == ncclUint64 || t == ncclInt64 || t == ncclFloat64 ? 8 : t == ncclUint8 || t == ncclInt8 ? 1 : 4; }

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    const uint64_t key = g_next++;
    g_worlds[key] = std::make_shared<World>();
    memset(id, 0, sizeof *id);
    memcpy(id, &key, sizeof key);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank)
{
    uint64_t key = 0;
    memcpy(&key, &id, sizeof key);
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_worlds.find(key);
        if (it == g_worlds.end()) return ncclInvalidArgument;
        w = it->second;
    }
    {
        std::lock_guard<std::mutex> lk(w->mu);
        if (w->world == 0) { w->world = world; w->ag.assign(world, nullptr); }
        else if (w->world != world) return ncclInvalidArgument;
    }
    *comm = reinterpret_cast<ncclComm_t>(new Comm{rank, w});
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *)
{
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    for (int r = 0; r < n; ++r) if (ncclResult_t e = ncclCommInitRank(&comms[r], n, id, r)) return e;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete reinterpret_cast<Comm *>(comm); return ncclSuccess; }

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) { *count = reinterpret_cast<const Comm *>(comm)->w->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) { *rank = reinterpret_cast<const Comm *>(comm)->rank; return ncclSuccess; }

ncclResult_t ncclGroupStart() { ++tl_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (tl_depth == 0) return ncclInvalidUsage;
    if (--tl_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(tl_ops);
    return run(ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (peer < 0 || peer >= c->w->world || peer == c->rank) return ncclInvalidArgument;
    return submit(Op{0, buf, nullptr, count * width(t), peer, c, s});
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (peer < 0 || peer >= c->w->world || peer == c->rank) return ncclInvalidArgument;
    return submit(Op{1, nullptr, buf, count * width(t), peer, c, s});
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    return submit(Op{2, send, recv, count * width(t), -1, c, s});
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "loopback_ccl: error"; }

}  // extern "C"
