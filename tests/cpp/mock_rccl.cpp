// mock_rccl.cpp -- TEST TRANSPORT, not RCCL: the nine entry points csrc/srt_comm.cpp resolves, implemented with HIP copies inside ONE
// process, so that the W > 1 branch of srt_render_frame_multi (gathered-buffer allocation, the grouped gather calls, the scatter from
// the rank-major buffer, the exchange timing) runs on a box with a single GPU.  Loaded through SRT_RCCL_LIB by
// tests/test_gpu_parity.py::test_comm_two_ranks_one_gpu_mock_transport; never by the product.
// A gather inside ncclGroupStart / ncclGroupEnd is recorded and carried out at ncclGroupEnd: the root's stream waits for an event on
// every sender's stream and copies count elements from the sender's buffer to recvbuf + rank * count -- the rank-major layout of
// ncclGather.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <vector>

// `shared`: a communicator formed by ncclCommInitRank with world > 1 -- its ranks live on different THREADS of this one process
// (tests/test_gpu_parity.py drives one rank per thread) and meet in the rendezvous below; ncclCommInitAll communicators are driven
// by one thread and use the grouped path.
struct MockComm { int rank, world; bool shared; };

// Rendezvous of the `world` ranks of the shared communicator (one at a time in a test process: every unique id is the same 128 bytes).
// A collective = every rank deposits its slot and the last one to arrive releases the others.
namespace {
struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    std::vector<const void *> send;
    std::vector<size_t> count;
    std::vector<hipEvent_t> ready;
    std::vector<std::vector<unsigned char>> host;      // all-gather: contributions staged on the host
};
Rendezvous g_rv;

// every rank calls this with its slot filled; returns when all `world` ranks of this collective have arrived.  `last` (out) is true
// for exactly one caller, which must call rv_release() after it has consumed the slots.
void rv_arrive(Rendezvous &rv, std::unique_lock<std::mutex> &lk, int world) {
    const unsigned long long gen = rv.generation;
    if (++rv.arrived == world) { rv.arrived = 0; rv.generation++; rv.cv.notify_all(); }
    else rv.cv.wait(lk, [&] { return rv.generation != gen; });
}
}  // namespace
struct PendingGather { const void *send; void *recv; size_t count; int root; MockComm *comm; hipStream_t stream; };
static thread_local std::vector<PendingGather> g_pending;
static thread_local int g_group_depth = 0;

static ncclResult_t flush() {
    void *recv = nullptr; hipStream_t root_stream = nullptr;
    for (const PendingGather &p : g_pending) if (p.comm->rank == p.root) { recv = p.recv; root_stream = p.stream; }
    if (!g_pending.empty() && !recv) return ncclInvalidArgument;
    for (const PendingGather &p : g_pending) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        if (hipEventRecord(ev, p.stream) != hipSuccess || hipStreamWaitEvent(root_stream, ev, 0) != hipSuccess) return ncclUnhandledCudaError;
        (void)hipEventDestroy(ev);
        if (hipMemcpyAsync((char *)recv + (size_t)p.comm->rank * p.count * sizeof(float), p.send, p.count * sizeof(float), hipMemcpyDeviceToDevice,
                           root_stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    g_pending.clear();
    return ncclSuccess;
}

extern "C" {
// identifies this library as the test transport: csrc/srt_comm.cpp honours SRT_COMM_TEST_SAME_DEVICE only when it finds this symbol
__attribute__((visibility("default"))) int srt_mock_rccl_marker = 1;
// all-gather of `count` elements per rank, used by the library at the start of every frame of a process-per-GPU communicator to agree
// on the exchange unit
__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
    if (!comm || (type != ncclUint32 && type != ncclFloat && type != ncclInt32)) return ncclInvalidArgument;
    MockComm *m = (MockComm *)comm;
    if (m->world == 1) return hipMemcpyAsync(recv, send, count * 4, hipMemcpyDeviceToDevice, stream) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
    if (!m->shared) return ncclInvalidUsage;      // (the library never all-gathers on a communicator one thread drives)
    // ranks on different threads: stage every contribution on the host, meet, hand everybody the whole table
    std::vector<unsigned char> mine(count * 4);
    if (hipMemcpyAsync(mine.data(), send, count * 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<unsigned char> all((size_t)m->world * count * 4);
    {
        std::unique_lock<std::mutex> lk(g_rv.mu);
        if ((int)g_rv.host.size() != m->world) g_rv.host.assign(m->world, {});
        g_rv.host[m->rank] = mine;
        rv_arrive(g_rv, lk, m->world);                 // everybody has deposited
        for (int r = 0; r < m->world; r++) memcpy(all.data() + (size_t)r * count * 4, g_rv.host[r].data(), count * 4);
        rv_arrive(g_rv, lk, m->world);                 // everybody has read: the slots may be reused
    }
    if (hipMemcpyAsync(recv, all.data(), all.size(), hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 7, sizeof(*id)); return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId, int rank) {
    *comm = (ncclComm_t) new MockComm{rank, world, world > 1};
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *) {
    for (int i = 0; i < n; i++) comms[i] = (ncclComm_t) new MockComm{i, n, false};
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete (MockComm *)comm; return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclGather(const void *send, void *recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm,
                                                              hipStream_t stream) {
    if (type != ncclFloat || !comm) return ncclInvalidArgument;
    MockComm *m = (MockComm *)comm;
    if (m->shared) {
        // ranks on different threads (all on one device in the tests): every rank records an event behind its send buffer, they meet,
        // and the root's stream waits for the events and copies the slots into the rank-major receive buffer
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, stream) != hipSuccess) return ncclUnhandledCudaError;
        ncclResult_t res = ncclSuccess;
        {
            std::unique_lock<std::mutex> lk(g_rv.mu);
            if ((int)g_rv.send.size() != m->world) { g_rv.send.assign(m->world, nullptr); g_rv.count.assign(m->world, 0); g_rv.ready.assign(m->world, nullptr); }
            g_rv.send[m->rank] = send; g_rv.count[m->rank] = count; g_rv.ready[m->rank] = ev;
            rv_arrive(g_rv, lk, m->world);
            if (m->rank == root) {
                for (int r = 0; r < m->world && res == ncclSuccess; r++) {
                    if (g_rv.count[r] != count) { res = ncclInvalidArgument; break; }      // (real RCCL would hang or corrupt here)
                    if (hipStreamWaitEvent(stream, g_rv.ready[r], 0) != hipSuccess ||
                        hipMemcpyAsync((char *)recv + (size_t)r * count * sizeof(float), g_rv.send[r], count * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)
                        res = ncclUnhandledCudaError;
                }
            }
            rv_arrive(g_rv, lk, m->world);             // the root has enqueued its copies: events may go
        }
        (void)hipEventDestroy(ev);
        return res;
    }
    g_pending.push_back({send, recv, count, root, m, stream});
    return g_group_depth ? ncclSuccess : flush();
}
__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() { g_group_depth++; return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() { return --g_group_depth == 0 ? flush() : ncclSuccess; }
__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock transport error"; }
}
