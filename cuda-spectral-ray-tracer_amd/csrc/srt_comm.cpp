// srt_comm.cpp -- multi-GPU fan-out of the render path behind the C-ABI.
//
// The reference renders one chunk at a time on one GPU (render_manager::step, rendering/render_manager.cu:3-66).  Here a
// frame (or chunk) is cut into 8x8-pixel tiles, rank r of W renders the tiles t with t % W == r (srt_set_partition), and
// the only exchange is ONE gather of the compact tile buffers to rank 0 over RCCL (xGMI inside a node), followed by one
// scatter kernel into rank 0's block-linear framebuffer.  Pixels are independent given the per-pixel seed
// 1984 + block-linear idx (rendering.cu:137), so the image is bit-identical for any W.
//
// Two ways to form a communicator:
//   srt_comm_init_all(devices, n)     one process drives n GPUs (ncclCommInitAll, one HIP stream per device)
//   srt_comm_init_rank(ctx, id, r, W) one process per GPU (torch.distributed.run / mpirun); the id comes from
//                                     srt_comm_unique_id on rank 0 and travels by whatever channel the launcher has
// RCCL is loaded with dlopen on first use: a single-GPU user of libsrt_hip.so never loads it, and a host process that
// carries its own RCCL copy (PyTorch does) is not disturbed.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "srt_host.h"
#include "srt_internal.h"

using namespace srt;

namespace {

struct RcclApi {
    void *handle = nullptr;
    bool preloaded = false;            // the process had an RCCL mapped already (e.g. PyTorch's) and we use that one
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    bool test_transport = false;       // the loaded library exports srt_mock_rccl_marker: tests/cpp/mock_rccl.cpp, not RCCL
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // A host process may already carry an RCCL (PyTorch maps its own copy): use THAT one -- two copies of the library in
        // one process would each keep their own bootstrap state and device-side buffers.  RTLD_NOLOAD only succeeds for an
        // image that is already mapped.  Otherwise load the system library.  SRT_RCCL_LIB names one explicit library instead
        // (no fall-backs: also how the CPU suite tests the "RCCL missing" path).
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::string first_error;
        // a candidate is accepted only if it exports everything this file calls (ncclGather is an RCCL extension: a copy the
        // host process carries may be older than the system's)
        auto accept = [&](void *h, bool preloaded) {
            if (!h || api.handle) { return; }
            bool ok = true;
            auto sym = [&](const char *name) { void *p = dlsym(h, name); if (!p) { ok = false; if (first_error.empty()) first_error = std::string("RCCL symbol missing: ") + name; } return p; };
            RcclApi cand;
            cand.GetUniqueId = (decltype(cand.GetUniqueId))sym("ncclGetUniqueId");
            cand.CommInitRank = (decltype(cand.CommInitRank))sym("ncclCommInitRank");
            cand.CommInitAll = (decltype(cand.CommInitAll))sym("ncclCommInitAll");
            cand.CommDestroy = (decltype(cand.CommDestroy))sym("ncclCommDestroy");
            cand.Gather = (decltype(cand.Gather))sym("ncclGather");
            cand.AllGather = (decltype(cand.AllGather))sym("ncclAllGather");
            cand.GroupStart = (decltype(cand.GroupStart))sym("ncclGroupStart");
            cand.GroupEnd = (decltype(cand.GroupEnd))sym("ncclGroupEnd");
            cand.GetErrorString = (decltype(cand.GetErrorString))sym("ncclGetErrorString");
            if (!ok) { dlclose(h); return; }
            cand.handle = h; cand.preloaded = preloaded;
            cand.test_transport = dlsym(h, "srt_mock_rccl_marker") != nullptr;
            api = cand;
        };
        auto try_open = [&](const char *n, int flags, bool preloaded) {
            if (api.handle) return;
            (void)dlerror();
            void *h = dlopen(n, flags);
            if (!h && !preloaded && first_error.empty()) { const char *e = dlerror(); first_error = e ? e : (std::string(n) + " not found"); }
            accept(h, preloaded);
        };
        if (const char *forced = getenv("SRT_RCCL_LIB")) {
            try_open(forced, RTLD_NOW | RTLD_LOCAL, false);
        } else {
            for (const char *n : names) try_open(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD, true);      // (RTLD_NOLOAD: only an image that is already mapped)
            for (const char *n : names) try_open(n, RTLD_NOW | RTLD_LOCAL, false);
        }
        if (!api.handle) api.error = "cannot load RCCL: " + (first_error.empty() ? std::string("librccl.so.1 not found") : first_error);
    });
    return api;
}

}  // namespace

struct srt_comm {
    uint32_t world = 1;
    std::vector<srt_ctx *> ctx;        // local contexts
    std::vector<uint32_t> rank;        // their global ranks
    std::vector<ncclComm_t> nccl;
    std::vector<hipStream_t> stream;   // one per local context
    bool owns_ctx = false;
    int root_local = -1;               // index of global rank 0 among the local contexts, -1 if it lives elsewhere
    float *d_gathered = nullptr;       // on rank 0's device: world * tiles_padded * 9 * 64 floats, rank-major
    size_t gathered_capacity = 0;
    uint32_t gather_planes = 3;        // 3: the quantised framebuffer only (12 B / pixel); 9: + the parity planes
    // one process per GPU only: the ranks set their plane count independently (srt_comm_set_gather_planes, or srt_set_gather_planes on
    // the caller-owned context), and ncclGather with different counts hangs or corrupts.  EVERY frame of such a communicator therefore
    // starts with the same collective on every rank -- a 4-byte all-gather of the count the rank's context will really use -- and every
    // rank fails with the same message when they differ; no rank ever enters ncclGather alone.
    uint32_t *d_agree = nullptr;       // world words on this rank's device
    std::vector<hipEvent_t> ev_g0, ev_g1;   // per local context: around the gather (+ scatter on rank 0) of the last frame
    bool gather_timed = false;
    std::string err;
};

namespace {

int cfail(srt_comm *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    set_global_error(msg);
    return code;
}
#define COMM_HIP(c, expr)                                                                         \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return cfail(c, SRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
#define COMM_NCCL(c, expr)                                                                        \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess) return cfail(c, SRT_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(_r)); \
    } while (0)

int device_of(srt_ctx *ctx) { return srt_ctx_device(ctx); }

}  // namespace

extern "C" {

int srt_comm_unique_id(unsigned char id[SRT_COMM_ID_BYTES]) {
    static_assert(SRT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id) return cfail(nullptr, SRT_ERR_INVALID, "srt_comm_unique_id: null id");
    RcclApi &R = rccl();
    if (!R.handle) return cfail(nullptr, SRT_ERR_UNSUPPORTED, "srt_comm_unique_id: " + R.error);
    ncclUniqueId u;
    COMM_NCCL(nullptr, R.GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return SRT_OK;
}

// SRT_OK when an RCCL is loadable in this process (cheap; lets a launcher agree on the exchange path BEFORE any rank enters
// the collective ncclCommInitRank -- a rank that failed earlier would leave the others waiting in the bootstrap)
int srt_comm_available(void) {
    RcclApi &R = rccl();
    if (!R.handle) return cfail(nullptr, SRT_ERR_UNSUPPORTED, "srt_comm_available: " + R.error);
    return SRT_OK;
}

int srt_comm_set_gather_planes(srt_comm *c, uint32_t planes) {
    if (!c || (planes != 3 && planes != 9)) return cfail(c, SRT_ERR_INVALID, "srt_comm_set_gather_planes: planes must be 3 or 9");
    for (srt_ctx *x : c->ctx) { int rc = srt_set_gather_planes(x, planes); if (rc != SRT_OK) return cfail(c, rc, srt_last_error(x)); }
    c->gather_planes = planes;
    return SRT_OK;
}

int srt_comm_init_rank(srt_ctx *ctx, const unsigned char id[SRT_COMM_ID_BYTES], uint32_t rank, uint32_t world, srt_comm **out) {
    if (!ctx || !id || !out || world == 0 || rank >= world) return cfail(nullptr, SRT_ERR_INVALID, "srt_comm_init_rank: bad argument");
    *out = nullptr;
    RcclApi &R = rccl();
    if (!R.handle) return cfail(nullptr, SRT_ERR_UNSUPPORTED, "srt_comm_init_rank: " + R.error);
    COMM_HIP(nullptr, hipSetDevice(device_of(ctx)));
    srt_comm *c = new srt_comm();
    c->world = world; c->ctx = {ctx}; c->rank = {rank}; c->owns_ctx = false; c->root_local = rank == 0 ? 0 : -1;
    c->nccl.assign(1, nullptr); c->stream.assign(1, nullptr); c->ev_g0.assign(1, nullptr); c->ev_g1.assign(1, nullptr);
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = R.CommInitRank(&c->nccl[0], (int)world, u, (int)rank);
    if (r != ncclSuccess) { int rc = cfail(nullptr, SRT_ERR_HIP, std::string("ncclCommInitRank: ") + R.GetErrorString(r)); delete c; return rc; }
    hipError_t e = hipStreamCreateWithFlags(&c->stream[0], hipStreamNonBlocking);
    if (e != hipSuccess) { int rc = cfail(nullptr, SRT_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); srt_comm_destroy(c); return rc; }
    int rc = srt_set_partition(ctx, rank, world);
    if (rc == SRT_OK) rc = srt_set_gather_planes(ctx, c->gather_planes);
    if (rc == SRT_OK && (hipEventCreate(&c->ev_g0[0]) != hipSuccess || hipEventCreate(&c->ev_g1[0]) != hipSuccess)) rc = cfail(nullptr, SRT_ERR_HIP, "srt_comm_init_rank: hipEventCreate failed");
    if (rc != SRT_OK) { srt_comm_destroy(c); return rc; }
    *out = c;
    return SRT_OK;
}

int srt_comm_init_all(const int *devices, int n, srt_comm **out) {
    if (!devices || n <= 0 || !out) return cfail(nullptr, SRT_ERR_INVALID, "srt_comm_init_all: bad argument");
    *out = nullptr;
    RcclApi &R = rccl();
    if (!R.handle) return cfail(nullptr, SRT_ERR_UNSUPPORTED, "srt_comm_init_all: " + R.error);
    // One rank per GPU.  Test hook: SRT_COMM_TEST_SAME_DEVICE=1 (with SRT_TEST_KNOBS=1) lets several ranks share a device, and is honoured
    // ONLY when the loaded transport identifies itself as the test transport (tests/cpp/mock_rccl.cpp exports srt_mock_rccl_marker; loaded
    // through SRT_RCCL_LIB) -- a duplicate device never reaches the real ncclCommInitAll.
    const char *same = getenv("SRT_COMM_TEST_SAME_DEVICE"), *knobs = getenv("SRT_TEST_KNOBS");
    const bool allow_same = same && same[0] == '1' && knobs && atoi(knobs) == 1 && R.test_transport;
    for (int i = 0; i < n && !allow_same; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) return cfail(nullptr, SRT_ERR_INVALID, "srt_comm_init_all: a device is listed twice (one rank per GPU)");
    srt_comm *c = new srt_comm();
    c->world = (uint32_t)n; c->owns_ctx = true; c->root_local = 0;
    c->ctx.assign(n, nullptr); c->rank.resize(n); c->nccl.assign(n, nullptr); c->stream.assign(n, nullptr);
    c->ev_g0.assign(n, nullptr); c->ev_g1.assign(n, nullptr);
    for (int i = 0; i < n; i++) {
        c->rank[i] = (uint32_t)i;
        int rc = srt_create(devices[i], &c->ctx[i]);
        if (rc == SRT_OK) rc = srt_set_partition(c->ctx[i], (uint32_t)i, (uint32_t)n);
        if (rc == SRT_OK) rc = srt_set_gather_planes(c->ctx[i], c->gather_planes);
        if (rc == SRT_OK && (hipEventCreate(&c->ev_g0[i]) != hipSuccess || hipEventCreate(&c->ev_g1[i]) != hipSuccess)) rc = cfail(nullptr, SRT_ERR_HIP, "srt_comm_init_all: hipEventCreate failed");
        if (rc == SRT_OK && hipStreamCreateWithFlags(&c->stream[i], hipStreamNonBlocking) != hipSuccess) rc = cfail(nullptr, SRT_ERR_HIP, "srt_comm_init_all: hipStreamCreate failed");
        if (rc != SRT_OK) { srt_comm_destroy(c); return rc; }
    }
    ncclResult_t r = R.CommInitAll(c->nccl.data(), n, devices);
    if (r != ncclSuccess) { int rc = cfail(nullptr, SRT_ERR_HIP, std::string("ncclCommInitAll: ") + R.GetErrorString(r)); srt_comm_destroy(c); return rc; }
    *out = c;
    return SRT_OK;
}

void srt_comm_destroy(srt_comm *c) {
    if (!c) return;
    RcclApi &R = rccl();
    for (size_t i = 0; i < c->ctx.size(); i++) {
        if (c->ctx[i]) { (void)hipSetDevice(device_of(c->ctx[i])); (void)hipDeviceSynchronize(); }
        if (c->nccl[i] && R.handle) (void)R.CommDestroy(c->nccl[i]);
        if (c->stream[i]) (void)hipStreamDestroy(c->stream[i]);
        if (i < c->ev_g0.size() && c->ev_g0[i]) (void)hipEventDestroy(c->ev_g0[i]);
        if (i < c->ev_g1.size() && c->ev_g1[i]) (void)hipEventDestroy(c->ev_g1[i]);
    }
    if (c->d_gathered && c->root_local >= 0) { (void)hipSetDevice(device_of(c->ctx[c->root_local])); (void)hipFree(c->d_gathered); }
    if (c->d_agree && !c->ctx.empty() && c->ctx[0]) { (void)hipSetDevice(device_of(c->ctx[0])); (void)hipFree(c->d_agree); }
    if (c->owns_ctx) for (srt_ctx *x : c->ctx) if (x) srt_destroy(x);
    delete c;
}

const char *srt_comm_last_error(const srt_comm *c) { return c ? c->err.c_str() : global_error(); }
uint32_t srt_comm_world(const srt_comm *c) { return c ? c->world : 0; }
uint32_t srt_comm_local_count(const srt_comm *c) { return c ? (uint32_t)c->ctx.size() : 0; }
srt_ctx *srt_comm_ctx(srt_comm *c, uint32_t local_index) { return (c && local_index < c->ctx.size()) ? c->ctx[local_index] : nullptr; }
srt_ctx *srt_comm_root_ctx(srt_comm *c) { return (c && c->root_local >= 0) ? c->ctx[c->root_local] : nullptr; }

int srt_comm_upload_scene(srt_comm *c, const srt_scene *s) {
    if (!c || !s) return cfail(c, SRT_ERR_INVALID, "srt_comm_upload_scene: null argument");
    for (srt_ctx *x : c->ctx) { int rc = srt_upload_scene(x, s); if (rc != SRT_OK) return cfail(c, rc, srt_last_error(x)); }
    return SRT_OK;
}
int srt_comm_set_camera(srt_comm *c, const srt_camera_data *cam) {
    if (!c || !cam) return cfail(c, SRT_ERR_INVALID, "srt_comm_set_camera: null argument");
    for (srt_ctx *x : c->ctx) { int rc = srt_set_camera(x, cam); if (rc != SRT_OK) return cfail(c, rc, srt_last_error(x)); }
    return SRT_OK;
}
int srt_comm_init_device_params(srt_comm *c, uint32_t tx, uint32_t ty, uint32_t bx, uint32_t by, uint32_t chunk_w, uint32_t chunk_h,
                                uint32_t spp, uint32_t bounce_limit, uint64_t seed) {
    if (!c) return cfail(c, SRT_ERR_INVALID, "srt_comm_init_device_params: null comm");
    // enqueue the re-seeding on every local device first, then wait for all of them: N devices take the time of one
    for (srt_ctx *x : c->ctx) {
        int rc = srt_internal_init_device_params(x, tx, ty, bx, by, chunk_w, chunk_h, spp, bounce_limit, seed, c->ctx.size() == 1 ? 1 : 0);
        if (rc != SRT_OK) return cfail(c, rc, srt_last_error(x));
    }
    if (c->ctx.size() > 1)
        for (srt_ctx *x : c->ctx) {
            COMM_HIP(c, hipSetDevice(device_of(x)));
            COMM_HIP(c, hipDeviceSynchronize());
        }
    return SRT_OK;
}

// render_manager::step's device half for W GPUs: every local rank renders its tiles on its own stream, ONE gather brings
// the compact tile buffers to rank 0, rank 0 scatters them into its block-linear framebuffer.  Asynchronous: returns once
// everything is enqueued; srt_comm_synchronize waits.
int srt_render_frame_multi(srt_comm *c, uint32_t width, uint32_t height, uint32_t offx, uint32_t offy) {
    if (!c) return cfail(c, SRT_ERR_INVALID, "srt_render_frame_multi: null comm");
    RcclApi &R = rccl();
    const size_t n_local = c->ctx.size();
    if (c->world > 1 && !c->owns_ctx) {
        // process-per-GPU communicator: agree on the exchange unit BEFORE anything of this frame is enqueued (see srt_comm::d_agree).
        // The count is the one the context's launch and srt_tile_buffer will use, whichever call set it.
        COMM_HIP(c, hipSetDevice(device_of(c->ctx[0])));
        if (!c->d_agree) COMM_HIP(c, hipMalloc((void **)&c->d_agree, (size_t)(c->world + 1) * sizeof(uint32_t)));
        const uint32_t mine = srt_internal_gather_planes(c->ctx[0]);
        COMM_HIP(c, hipMemcpyAsync(c->d_agree + c->world, &mine, sizeof(mine), hipMemcpyHostToDevice, c->stream[0]));
        COMM_NCCL(c, R.AllGather(c->d_agree + c->world, c->d_agree, 1, ncclUint32, c->nccl[0], c->stream[0]));
        std::vector<uint32_t> all(c->world);
        COMM_HIP(c, hipMemcpyAsync(all.data(), c->d_agree, c->world * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream[0]));
        COMM_HIP(c, hipStreamSynchronize(c->stream[0]));
        for (uint32_t r = 0; r < c->world; r++)
            if (all[r] != all[0])
                return cfail(c, SRT_ERR_INVALID, "srt_render_frame_multi: the ranks disagree on the exchange unit (rank 0 gathers " + std::to_string(all[0]) +
                                                     " planes, rank " + std::to_string(r) + " " + std::to_string(all[r]) + ", this rank " + std::to_string(mine) +
                                                     "): set the same value with srt_comm_set_gather_planes / srt_set_gather_planes on every rank");
        c->gather_planes = mine;
    }
    for (size_t i = 0; i < n_local; i++) {
        int rc = srt_render_chunk(c->ctx[i], width, height, offx, offy, c->stream[i]);
        if (rc != SRT_OK) return cfail(c, rc, srt_last_error(c->ctx[i]));
    }
    // every rank's exchange unit has the same size: tiles_padded * gather_planes * 64 floats (the first plane groups of its buffer)
    void *tiles0 = nullptr; size_t n_floats = 0; uint32_t tl = 0, tp = 0;
    int rc = srt_tile_buffer(c->ctx[0], &tiles0, &n_floats, &tl, &tp);
    if (rc != SRT_OK) return cfail(c, rc, srt_last_error(c->ctx[0]));
    if (c->world == 1) {
        rc = srt_scatter_tiles(c->ctx[0], nullptr, c->stream[0]);
        return rc == SRT_OK ? SRT_OK : cfail(c, rc, srt_last_error(c->ctx[0]));
    }
    if (c->root_local >= 0 && (size_t)c->world * n_floats > c->gathered_capacity) {
        COMM_HIP(c, hipSetDevice(device_of(c->ctx[c->root_local])));
        COMM_HIP(c, hipStreamSynchronize(c->stream[c->root_local]));   // a scatter of the previous frame may still read it
        if (c->d_gathered) { (void)hipFree(c->d_gathered); c->d_gathered = nullptr; c->gathered_capacity = 0; }
        COMM_HIP(c, hipMalloc((void **)&c->d_gathered, (size_t)c->world * n_floats * sizeof(float)));
        c->gathered_capacity = (size_t)c->world * n_floats;
    }
    for (size_t i = 0; i < n_local; i++) {
        COMM_HIP(c, hipSetDevice(device_of(c->ctx[i])));
        COMM_HIP(c, hipEventRecord(c->ev_g0[i], c->stream[i]));
    }
    COMM_NCCL(c, R.GroupStart());
    for (size_t i = 0; i < n_local; i++) {
        void *tiles = nullptr; size_t nf = 0;
        rc = srt_tile_buffer(c->ctx[i], &tiles, &nf, nullptr, nullptr);
        if (rc != SRT_OK || nf != n_floats) { (void)R.GroupEnd(); return cfail(c, SRT_ERR_INVALID, "srt_render_frame_multi: tile buffers of the ranks differ in size"); }
        ncclResult_t r = R.Gather(tiles, (int)i == c->root_local ? c->d_gathered : nullptr, n_floats, ncclFloat, 0, c->nccl[i], c->stream[i]);
        if (r != ncclSuccess) { (void)R.GroupEnd(); return cfail(c, SRT_ERR_HIP, std::string("ncclGather: ") + R.GetErrorString(r)); }
    }
    COMM_NCCL(c, R.GroupEnd());
    if (c->root_local >= 0) {
        rc = srt_scatter_tiles(c->ctx[c->root_local], c->d_gathered, c->stream[c->root_local]);
        if (rc != SRT_OK) return cfail(c, rc, srt_last_error(c->ctx[c->root_local]));
    }
    for (size_t i = 0; i < n_local; i++) {
        COMM_HIP(c, hipSetDevice(device_of(c->ctx[i])));
        COMM_HIP(c, hipEventRecord(c->ev_g1[i], c->stream[i]));
    }
    c->gather_timed = true;
    return SRT_OK;
}

// time the local ranks spent between the end of their render kernel and the end of the exchange of the last frame (the gather,
// plus the scatter on rank 0; includes waiting for the slowest rank): max over the local contexts, ms.  0 for a 1-rank world.
int srt_comm_last_gather_ms(srt_comm *c, float *ms) {
    if (!c || !ms) return cfail(c, SRT_ERR_INVALID, "srt_comm_last_gather_ms: null argument");
    *ms = 0.f;
    if (c->world == 1 || !c->gather_timed) return SRT_OK;
    for (size_t i = 0; i < c->ctx.size(); i++) {
        float t = 0.f;
        COMM_HIP(c, hipSetDevice(device_of(c->ctx[i])));
        COMM_HIP(c, hipEventSynchronize(c->ev_g1[i]));
        COMM_HIP(c, hipEventElapsedTime(&t, c->ev_g0[i], c->ev_g1[i]));
        *ms = t > *ms ? t : *ms;
    }
    return SRT_OK;
}

int srt_comm_synchronize(srt_comm *c) {
    if (!c) return cfail(c, SRT_ERR_INVALID, "srt_comm_synchronize: null comm");
    for (size_t i = 0; i < c->ctx.size(); i++) {
        COMM_HIP(c, hipSetDevice(device_of(c->ctx[i])));
        COMM_HIP(c, hipStreamSynchronize(c->stream[i]));
    }
    return SRT_OK;
}

// closest-hit queries of the last frame over the local ranks, and the slowest local render kernel (ms)
int srt_comm_stats(srt_comm *c, uint64_t *rays, uint64_t *paths, float *max_kernel_ms) {
    if (!c) return cfail(c, SRT_ERR_INVALID, "srt_comm_stats: null comm");
    uint64_t r = 0, p = 0; float ms = 0.f;
    for (srt_ctx *x : c->ctx) {
        srt_stats st; float k = 0.f;
        int rc = srt_get_stats(x, &st);
        if (rc == SRT_OK) rc = srt_last_kernel_ms(x, &k);
        if (rc != SRT_OK) return cfail(c, rc, srt_last_error(x));
        r += st.rays; p += st.paths; ms = k > ms ? k : ms;
    }
    if (rays) *rays = r;
    if (paths) *paths = p;
    if (max_kernel_ms) *max_kernel_ms = ms;
    return SRT_OK;
}

}  // extern "C"
