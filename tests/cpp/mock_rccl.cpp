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

#include <vector>

struct MockComm { int rank, world; };
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
// all-gather of `count` elements per rank, used by the library once per communicator to agree on the exchange unit; in this
// single-process transport every rank of a communicator formed by ncclCommInitRank is alone in its process (world 1 in the tests)
// or shares the process (ncclCommInitAll: the library never calls it there): copy own contribution to every slot it owns
__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
    if (!comm || (type != ncclUint32 && type != ncclFloat && type != ncclInt32)) return ncclInvalidArgument;
    MockComm *m = (MockComm *)comm;
    if (m->world != 1) return ncclInvalidUsage;      // (would need a rendezvous between ranks: not what this mock is for)
    return hipMemcpyAsync(recv, send, count * 4, hipMemcpyDeviceToDevice, stream) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 7, sizeof(*id)); return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId, int rank) {
    *comm = (ncclComm_t) new MockComm{rank, world};
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *) {
    for (int i = 0; i < n; i++) comms[i] = (ncclComm_t) new MockComm{i, n};
    return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete (MockComm *)comm; return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclGather(const void *send, void *recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm,
                                                              hipStream_t stream) {
    if (type != ncclFloat || !comm) return ncclInvalidArgument;
    g_pending.push_back({send, recv, count, root, (MockComm *)comm, stream});
    return g_group_depth ? ncclSuccess : flush();
}
__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() { g_group_depth++; return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() { return --g_group_depth == 0 ? flush() : ncclSuccess; }
__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock transport error"; }
}
