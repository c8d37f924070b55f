// A TEST DOUBLE for the dozen RCCL entry points csrc/group.hip uses — NOT RCCL.  Every "rank" lives in ONE process on
// ONE device (a test box has a single GPU and RCCL refuses two ranks on one device), so a transfer between ranks is a
// stream-ordered device-to-device copy and a collective is a small kernel; what is kept of the real thing is the
// CONTRACT the library programs against: communicators made by ncclCommInitAll or by a blocking ncclCommInitRank
// rendezvous on a unique id, operations queued between ncclGroupStart / ncclGroupEnd, a send matched with the peer's
// receive of the same size in posting order, collectives matched by their order on the communicator, everything
// ordered against the stream it was issued on.  A mismatch the real library would hang or corrupt on (sizes that
// differ, a collective one rank skips) is an error or a timeout here.  Loaded through GSPLAT_RCCL_LIB by
// tests/_fake_rccl_session.py, which renders frames through gsplat_group_* with two and three members — the one-thread
// local form and one thread per rank — and compares every member's frame with the single-GPU one.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace {

constexpr int MAX_RANKS = 8;
constexpr auto PATIENCE = std::chrono::seconds(20);  // a peer that never posts its half: error, not a hang

struct SendPost {
    const void *buf = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr;    // recorded on the sender's stream when the data is final
    hipEvent_t copied = nullptr;   // recorded on the receiver's stream behind the copy
    bool consumed = false;
};
struct Contribution {
    const void *send = nullptr;
    void *recv = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;
    bool present = false;
};
struct Collective {
    int kind = 0, root = 0;        // 1 all-reduce(max, u32), 2 broadcast
    size_t bytes = 0;
    Contribution c[MAX_RANKS];
    int arrived = 0, left = 0;
    bool done = false, failed = false;
    hipEvent_t finished = nullptr;
};
struct World {
    int n = 0, joined = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<SendPost>>> mail;  // (src, dst) in posting order
    std::map<uint64_t, Collective> colls;                                       // by position on the communicator
};

std::mutex g_mutex;
std::map<std::string, std::shared_ptr<World>> g_worlds;  // by unique id
uint64_t g_next_id = 1;

struct Op {
    int kind = 0;  // 1 all-reduce, 2 broadcast, 3 send, 4 recv
    const void *send = nullptr;
    void *recv = nullptr;
    size_t bytes = 0;
    int peer = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    uint64_t seq = 0;
    std::shared_ptr<SendPost> post;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

struct MaxArgs {
    const uint32_t *src[MAX_RANKS];
    uint32_t *dst[MAX_RANKS];
    int n, words;
};
__global__ void max_words_kernel(MaxArgs a) {  // word by word (one lane per word: a batch's frames each have their own)
    if ((int)threadIdx.x >= a.words) return;
    uint32_t m = 0;
    for (int r = 0; r < a.n; ++r) m = max(m, a.src[r][threadIdx.x]);
    for (int r = 0; r < a.n; ++r) a.dst[r][threadIdx.x] = m;
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

}  // namespace

struct ncclComm {
    std::shared_ptr<World> w;
    int rank = 0;
    uint64_t coll_seq = 0;
};

namespace {

#define HIP_OK(expr) do { if ((expr) != hipSuccess) return ncclUnhandledCudaError; } while (0)

// phase 1: make this rank's half of the operation visible to its peers (never blocks)
ncclResult_t post(Op &op) {
    World &w = *op.comm->w;
    hipEvent_t ready = nullptr;
    HIP_OK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    HIP_OK(hipEventRecord(ready, op.stream));
    std::unique_lock<std::mutex> lock(w.m);
    if (op.kind == 3) {
        op.post = std::make_shared<SendPost>();
        op.post->buf = op.send; op.post->bytes = op.bytes; op.post->ready = ready;
        w.mail[{op.comm->rank, op.peer}].push_back(op.post);
    } else if (op.kind == 4) {
        (void)hipEventDestroy(ready);  // (a receive has nothing to announce)
    } else {
        Collective &c = w.colls[op.seq];
        if (c.arrived == 0) { c.kind = op.kind; c.root = op.peer; c.bytes = op.bytes; }
        if (c.kind != op.kind || c.bytes != op.bytes || (op.kind == 2 && c.root != op.peer) || c.c[op.comm->rank].present)
            c.failed = true;  // ranks disagree on what collective number `seq` is
        Contribution &mine = c.c[op.comm->rank];
        mine.send = op.send; mine.recv = op.recv; mine.stream = op.stream; mine.ready = ready; mine.present = true;
        if (++c.arrived == w.n && !c.failed) {  // the last to arrive does the work, on its own stream
            bool ok = true;
            for (int r = 0; r < w.n; ++r) ok = ok && hipStreamWaitEvent(op.stream, c.c[r].ready, 0) == hipSuccess;
            if (c.kind == 1) {
                MaxArgs a;
                a.n = w.n;
                a.words = (int)(c.bytes / 4);
                for (int r = 0; r < w.n; ++r) { a.src[r] = static_cast<const uint32_t *>(c.c[r].send); a.dst[r] = static_cast<uint32_t *>(c.c[r].recv); }
                hipLaunchKernelGGL(max_words_kernel, dim3(1), dim3(64), 0, op.stream, a);
                ok = ok && hipGetLastError() == hipSuccess;
            } else {
                const void *src = c.c[c.root].send;
                for (int r = 0; r < w.n; ++r)
                    if (c.c[r].recv != src)
                        ok = ok && hipMemcpyAsync(c.c[r].recv, src, c.bytes, hipMemcpyDeviceToDevice, op.stream) == hipSuccess;
            }
            ok = ok && hipEventCreateWithFlags(&c.finished, hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventRecord(c.finished, op.stream) == hipSuccess;
            c.failed = !ok;
            c.done = true;
        }
    }
    lock.unlock();
    w.cv.notify_all();
    return ncclSuccess;
}

// phase 2: take what the peers posted (blocks until they have)
ncclResult_t complete(Op &op) {
    World &w = *op.comm->w;
    std::unique_lock<std::mutex> lock(w.m);
    if (op.kind == 4) {
        auto &queue = w.mail[{op.peer, op.comm->rank}];
        if (!w.cv.wait_for(lock, PATIENCE, [&] { return !queue.empty(); })) return ncclInternalError;  // no matching send
        std::shared_ptr<SendPost> p = queue.front();
        queue.pop_front();
        if (p->bytes != op.bytes) { p->consumed = true; w.cv.notify_all(); return ncclInvalidArgument; }
        HIP_OK(hipStreamWaitEvent(op.stream, p->ready, 0));
        HIP_OK(hipMemcpyAsync(op.recv, p->buf, op.bytes, hipMemcpyDeviceToDevice, op.stream));
        HIP_OK(hipEventCreateWithFlags(&p->copied, hipEventDisableTiming));
        HIP_OK(hipEventRecord(p->copied, op.stream));
        p->consumed = true;
        lock.unlock();
        w.cv.notify_all();
    } else if (op.kind == 1 || op.kind == 2) {
        Collective &c = w.colls[op.seq];
        if (!w.cv.wait_for(lock, PATIENCE, [&] { return c.done || c.failed; })) return ncclInternalError;  // a rank skipped it
        const bool failed = c.failed;
        if (!failed) HIP_OK(hipStreamWaitEvent(op.stream, c.finished, 0));
        (void)hipEventDestroy(c.c[op.comm->rank].ready);
        if (++c.left == w.n) {
            if (c.finished) (void)hipEventDestroy(c.finished);
            w.colls.erase(op.seq);
        }
        if (failed) return ncclInvalidUsage;
    }
    return ncclSuccess;
}

// phase 3: a sender's buffer may be rewritten by later work on its stream only after the receiver's copy
ncclResult_t release(Op &op) {
    if (op.kind != 3) return ncclSuccess;
    World &w = *op.comm->w;
    std::unique_lock<std::mutex> lock(w.m);
    if (!w.cv.wait_for(lock, PATIENCE, [&] { return op.post->consumed; })) return ncclInternalError;  // no matching receive
    if (op.post->copied) {
        HIP_OK(hipStreamWaitEvent(op.stream, op.post->copied, 0));
        (void)hipEventDestroy(op.post->copied);
    }
    (void)hipEventDestroy(op.post->ready);
    return op.post->copied ? ncclSuccess : ncclInvalidArgument;
}

ncclResult_t run(std::vector<Op> &ops) {
    ncclResult_t first = ncclSuccess;
    auto note = [&](ncclResult_t r) { if (r != ncclSuccess && first == ncclSuccess) first = r; };
    for (Op &op : ops) note(post(op));
    for (Op &op : ops) note(complete(op));
    for (Op &op : ops) note(release(op));
    ops.clear();
    return first;
}

ncclResult_t enqueue(Op op) {
    if (!op.comm || op.comm->rank < 0) return ncclInvalidArgument;
    if (op.kind <= 2) op.seq = op.comm->coll_seq++;
    t_ops.push_back(op);
    if (t_depth == 0) return run(t_ops);
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lock(g_mutex);
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "fake-rccl-world-%llu", (unsigned long long)g_next_id++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        std::shared_ptr<World> &slot = g_worlds[std::string(id.internal, strnlen(id.internal, sizeof id.internal))];
        if (!slot) { slot = std::make_shared<World>(); slot->n = nranks; }
        w = slot;
    }
    if (w->n != nranks) return ncclInvalidArgument;
    std::unique_lock<std::mutex> lock(w->m);
    ++w->joined;
    w->cv.notify_all();
    if (!w->cv.wait_for(lock, PATIENCE, [&] { return w->joined >= w->n; })) return ncclInternalError;  // a rank never came
    ncclComm *c = new ncclComm();
    c->w = w; c->rank = rank;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int * /*devlist: one device here*/) {
    if (!comms || ndev < 1 || ndev > MAX_RANKS) return ncclInvalidArgument;
    auto w = std::make_shared<World>();
    w->n = w->joined = ndev;
    for (int r = 0; r < ndev; ++r) {
        comms[r] = new ncclComm();
        comms[r]->w = w; comms[r]->rank = r;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return run(t_ops);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    // the one all-reduce the library issues: MAX of one u32 per frame (1 word, or up to 64 for a batch of frames)
    if (type != ncclUint32 || op != ncclMax || count < 1 || count > 64) return ncclInvalidUsage;
    Op o; o.kind = 1; o.send = send; o.recv = recv; o.bytes = 4 * count; o.comm = comm; o.stream = stream;
    return enqueue(o);
}
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm,
                           hipStream_t stream) {
    Op o; o.kind = 2; o.send = send; o.recv = recv; o.bytes = count * type_bytes(type); o.peer = root; o.comm = comm; o.stream = stream;
    if (!comm || root < 0 || root >= comm->w->n) return ncclInvalidArgument;
    return enqueue(o);
}
ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    Op o; o.kind = 3; o.send = send; o.bytes = count * type_bytes(type); o.peer = peer; o.comm = comm; o.stream = stream;
    if (!comm || peer < 0 || peer >= comm->w->n || peer == comm->rank) return ncclInvalidArgument;
    return enqueue(o);
}
ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    Op o; o.kind = 4; o.recv = recv; o.bytes = count * type_bytes(type); o.peer = peer; o.comm = comm; o.stream = stream;
    if (!comm || peer < 0 || peer >= comm->w->n || peer == comm->rank) return ncclInvalidArgument;
    return enqueue(o);
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (fake rccl)";
        case ncclInvalidArgument: return "invalid argument (fake rccl: sizes of a send and its receive differ, or a bad rank)";
        case ncclInvalidUsage: return "invalid usage (fake rccl: the ranks disagree on a collective)";
        case ncclInternalError: return "a peer never posted its half (fake rccl: the real library would hang here)";
        default: return "error (fake rccl)";
    }
}

}  // extern "C"
