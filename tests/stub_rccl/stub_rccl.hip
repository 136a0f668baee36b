// stub_rccl.hip — a TEST DOUBLE for librccl, not a communication library.  Test infrastructure only (tests/test_gpu_stub_rccl.py).
//
// Why it exists: the collective path of the product — mpt_comm_create_all / mpt_comm_create_rank / mpt_reduce_sum with more than one
// rank (metalpathtracer_amd/csrc/mpt_hip.hip: ncclGroupStart -> N x ncclReduce -> ncclGroupEnd -> stream syncs, and the abort path) —
// cannot execute on the one-GPU boxes this project is built on: RCCL refuses two ranks on one device.  The product dlopen()s
// "librccl.so.1"; with this library first on LD_LIBRARY_PATH it binds the nine entry points below instead, and every line of that
// path runs: two contexts on GPU 0 render the two tile shards of one image, the "collective" lands the sum in the root's buffer.
// It is ORCHESTRATION evidence — argument order, group bracketing, which stream each rank's call is enqueued on, what happens after a
// rank aborts — and says nothing about xGMI, RCCL's kernels or scaling.
//
// Semantics kept from NCCL: ncclCommInitAll makes one clique of N communicators; ncclCommInitRank joins the clique named by the unique
// id (it does not block here: the test's ranks are threads of one process); ncclReduce must be posted by every rank of the clique with the
// same count / type / op / root, inside a group or on its own; the posts of a clique complete together: the call that completes the clique
// performs the reduction — root.recv = sum over ranks, in rank order, of rank.send — as device kernels on the ROOT's stream, behind an
// event recorded on every other rank's stream (their renders are finished before their buffers are read) and in front of an event each of
// them waits for (a rank's buffer is not reused before it has been read): stream-ordered, like the real call.  A rank that waits for its
// peers longer than STUB_RCCL_TIMEOUT_MS (default 20000) gets ncclSystemError.  ncclCommAbort marks the communicator AND wakes and fails the
// peers that wait in the clique (ncclRemoteError) — the behaviour the product's comm_abort() relies on.  Every call is counted
// (stub_rccl_counters) so that the test can assert what the product actually called.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared tests/stub_rccl/stub_rccl.hip -o tests/stub_rccl/_build/librccl.so.1
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Post {
    bool posted = false;
    const void* send = nullptr;
    void* recv = nullptr;
    size_t count = 0;
    int root = -1;
    hipStream_t stream = nullptr;
    int device = 0;
};
struct Clique {
    int n = 0;
    std::vector<Post> posts;
    std::vector<bool> joined, aborted;
    unsigned long long generation = 0;   // completed collectives
    ncclResult_t last = ncclSuccess;     // result of the collective that completed last
    std::condition_variable cv;
};
struct StubComm {
    std::shared_ptr<Clique> clique;
    int rank = 0, device = 0;
    bool aborted = false;
};
std::mutex g_mu;
std::map<std::string, std::shared_ptr<Clique>> g_by_id;   // cliques being assembled by ncclCommInitRank
unsigned long long g_counters[16];   // 0 GetUniqueId, 1 CommInitAll, 2 CommInitRank, 3 CommDestroy, 4 CommAbort, 5 Reduce, 6 GroupStart, 7 GroupEnd,
                                     // 8 collectives completed, 9 collectives failed, 10 floats added, 11 Reduce calls outside a group
thread_local int t_group_depth = 0;
thread_local std::vector<StubComm*> t_group_posts;   // communicators posted inside the open group of this thread
unsigned g_next_id = 1;

__global__ void k_stub_add(float* dst, const float* src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
__global__ void k_stub_copy(float* dst, const float* src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int timeout_ms() {
    const char* e = getenv("STUB_RCCL_TIMEOUT_MS");
    return e && atoi(e) > 0 ? atoi(e) : 20000;
}

// all ranks have posted (caller holds g_mu): root.recv = sum over ranks in rank order, stream-ordered as described above
ncclResult_t run_collective(Clique& q) {
    const Post& r0 = q.posts[0];
    for (int r = 1; r < q.n; ++r)
        if (q.posts[r].count != r0.count || q.posts[r].root != r0.root) return ncclInvalidArgument;
    const int root = r0.root;
    if (root < 0 || root >= q.n) return ncclInvalidArgument;
    const Post& R = q.posts[root];
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return ncclUnhandledCudaError;
    auto fail = [&](ncclResult_t e) {
        (void)hipSetDevice(prev);
        return e;
    };
    // the peers' streams must have produced their buffers before the root's stream reads them
    for (int r = 0; r < q.n; ++r) {
        if (r == root) continue;
        hipEvent_t ev;
        if (hipSetDevice(q.posts[r].device) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(ncclUnhandledCudaError);
        if (hipEventRecord(ev, q.posts[r].stream) != hipSuccess) return fail(ncclUnhandledCudaError);
        if (hipSetDevice(R.device) != hipSuccess || hipStreamWaitEvent(R.stream, ev, 0) != hipSuccess) return fail(ncclUnhandledCudaError);
        (void)hipEventDestroy(ev);   // (released once the wait has been satisfied)
    }
    if (hipSetDevice(R.device) != hipSuccess) return fail(ncclUnhandledCudaError);
    const size_t n = R.count;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    float* acc = (float*)R.recv;
    // in place at the root (what the product does: send == recv == the context's HDR sum): the root's own term is already in the
    // accumulator, the other ranks are added to it in rank order; out of place: acc = send[0], then the others in rank order
    const bool in_place = R.send == (const void*)acc;
    if (!in_place) hipLaunchKernelGGL(k_stub_copy, dim3(grid), dim3(256), 0, R.stream, acc, (const float*)q.posts[0].send, n);
    for (int r = in_place ? 0 : 1; r < q.n; ++r) {
        if (in_place && r == root) continue;
        hipLaunchKernelGGL(k_stub_add, dim3(grid), dim3(256), 0, R.stream, acc, (const float*)q.posts[r].send, n);
        g_counters[10] += n;
    }
    if (hipGetLastError() != hipSuccess) return fail(ncclUnhandledCudaError);
    // ... and no peer reuses its buffer before the root has read it
    hipEvent_t done;
    if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess || hipEventRecord(done, R.stream) != hipSuccess) return fail(ncclUnhandledCudaError);
    for (int r = 0; r < q.n; ++r) {
        if (r == root) continue;
        if (hipSetDevice(q.posts[r].device) != hipSuccess || hipStreamWaitEvent(q.posts[r].stream, done, 0) != hipSuccess) return fail(ncclUnhandledCudaError);
    }
    (void)hipEventDestroy(done);
    (void)hipSetDevice(prev);
    return ncclSuccess;
}

// post-processing of one rank's Reduce: completes the clique or waits for it (lock held on entry and exit)
ncclResult_t complete_or_wait(std::unique_lock<std::mutex>& lk, StubComm* c) {
    Clique& q = *c->clique;
    for (int r = 0; r < q.n; ++r)
        if (q.aborted[r]) {   // a peer gave up: this collective can never complete
            for (auto& p : q.posts) p = Post();
            q.last = ncclRemoteError;
            ++q.generation;
            ++g_counters[9];
            q.cv.notify_all();
            return ncclRemoteError;
        }
    bool all = true;
    for (int r = 0; r < q.n; ++r) all = all && q.posts[r].posted;
    if (all) {
        const ncclResult_t res = run_collective(q);
        for (auto& p : q.posts) p = Post();
        q.last = res;
        ++q.generation;
        ++g_counters[res == ncclSuccess ? 8 : 9];
        q.cv.notify_all();
        return res;
    }
    const unsigned long long gen = q.generation;
    const bool ok = q.cv.wait_for(lk, std::chrono::milliseconds(timeout_ms()), [&] { return q.generation != gen; });
    if (!ok) {
        q.posts[c->rank] = Post();
        ++g_counters[9];
        return ncclSystemError;   // the peers never came
    }
    return q.last;
}
}  // namespace

extern "C" {
// (test hook, not an NCCL entry point)
void stub_rccl_counters(unsigned long long* out16) {
    std::lock_guard<std::mutex> lk(g_mu);
    memcpy(out16, g_counters, sizeof g_counters);
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_counters[0];
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "stub-rccl-%u", g_next_id++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devs) {
    if (!comms || n < 1) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_counters[1];
    auto q = std::make_shared<Clique>();
    q->n = n;
    q->posts.resize(n);
    q->joined.assign(n, true);
    q->aborted.assign(n, false);
    for (int r = 0; r < n; ++r) {
        StubComm* c = new StubComm();
        c->clique = q;
        c->rank = r;
        c->device = devs ? devs[r] : r;
        comms[r] = (ncclComm_t)c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_counters[2];
    const std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
    std::shared_ptr<Clique>& q = g_by_id[key];
    if (!q) {
        q = std::make_shared<Clique>();
        q->n = nranks;
        q->posts.resize(nranks);
        q->joined.assign(nranks, false);
        q->aborted.assign(nranks, false);
    }
    if (q->n != nranks || q->joined[rank]) return ncclInvalidArgument;
    q->joined[rank] = true;
    StubComm* c = new StubComm();
    c->clique = q;
    c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) c->device = 0;
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_counters[3];
    delete (StubComm*)comm;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_counters[4];
    StubComm* c = (StubComm*)comm;
    Clique& q = *c->clique;
    q.aborted[c->rank] = true;
    // peers that wait in the clique are released with an error
    bool waiting = false;
    for (auto& p : q.posts) waiting = waiting || p.posted;
    if (waiting) {
        for (auto& p : q.posts) p = Post();
        q.last = ncclRemoteError;
        ++q.generation;
        ++g_counters[9];
        q.cv.notify_all();
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_counters[6];
    ++t_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root, ncclComm_t comm,
                        hipStream_t stream) {
    if (!comm || !sendbuff || datatype != ncclFloat32 || op != ncclSum) return ncclInvalidArgument;   // (all the product uses)
    std::unique_lock<std::mutex> lk(g_mu);
    ++g_counters[5];
    StubComm* c = (StubComm*)comm;
    Clique& q = *c->clique;
    if (root < 0 || root >= q.n || (c->rank == root && !recvbuff)) return ncclInvalidArgument;
    Post& p = q.posts[c->rank];
    if (p.posted) return ncclInvalidUsage;   // two collectives of one communicator in flight
    p.posted = true;
    p.send = sendbuff;
    p.recv = recvbuff;
    p.count = count;
    p.root = root;
    p.stream = stream;
    p.device = c->device;
    if (t_group_depth > 0) {
        t_group_posts.push_back(c);
        return ncclSuccess;   // completes at ncclGroupEnd
    }
    ++g_counters[11];
    return complete_or_wait(lk, c);
}

ncclResult_t ncclGroupEnd() {
    std::unique_lock<std::mutex> lk(g_mu);
    ++g_counters[7];
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    std::vector<StubComm*> posts;
    posts.swap(t_group_posts);
    ncclResult_t res = ncclSuccess;
    // one completion per clique touched by this group (the posts of a clique made inside one group complete together)
    std::vector<Clique*> seen;
    for (StubComm* c : posts) {
        Clique* q = c->clique.get();
        bool dup = false;
        for (Clique* s : seen) dup = dup || s == q;
        if (dup) continue;
        seen.push_back(q);
        const ncclResult_t r = complete_or_wait(lk, c);
        if (res == ncclSuccess) res = r;
    }
    return res;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (stub rccl)";
        case ncclUnhandledCudaError: return "unhandled hip error (stub rccl)";
        case ncclSystemError: return "peers never joined the collective (stub rccl)";
        case ncclInvalidArgument: return "invalid argument (stub rccl)";
        case ncclInvalidUsage: return "invalid usage (stub rccl)";
        case ncclRemoteError: return "a peer aborted its communicator (stub rccl)";
        default: return "error (stub rccl)";
    }
}
}
