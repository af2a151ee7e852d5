// srt_capi.cpp -- device half of the C-ABI: one srt_ctx = one GPU's renderer
// (replaces `renderer`, rendering/rendering.cuh:39-155, and the device half of render_manager::step).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "srt_host.h"
#include "srt_internal.h"

using namespace srt;

#ifndef SRT_FRINGE_STRIDE_L2
#define SRT_FRINGE_STRIDE_L2 96
#endif
static constexpr uint32_t kFringeStrideL2 = SRT_FRINGE_STRIDE_L2;
// step-choice weights when the inner tree exceeds the LDS cache (256 = an INNER visit)
static constexpr uint32_t kScoreShadeL2 = 320u, kScoreFringeL2 = 800u;   // FRINGE record stride for trees that do not fit LDS

struct srt_ctx {
    int device = 0;
    std::string err;
    // scene images in HBM
    float *d_nodes = nullptr, *d_nodes_sw = nullptr, *d_fringe = nullptr, *d_tris = nullptr, *d_mat_sd = nullptr, *d_mat_par = nullptr, *d_shade = nullptr, *d_cmf = nullptr;
    int root_ref = 0, stack_depth = 1, n_inner = 0, n_records = 0;
    bool paired = false;               // the uploaded tree has no node with exactly one leaf child (srt_scene_is_paired)
    uint32_t fringe_stride = 96;       // bytes between FRINGE records in d_fringe
    uint32_t n_tris = 0;
    uint32_t n_materials = 0;
    bool scene_ready = false, camera_ready = false, params_ready = false;
    srt_camera_data cam;
    // launch geometry (renderer::init_device_params)
    uint32_t tx = 0, ty = 0, bx = 0, by = 0, chunk_w = 0, chunk_h = 0, spp = 0, bounce = 0;
    uint64_t seed = SRT_DEFAULT_SEED;
    uint32_t n_lanes = 0;
    uint32_t rank = 0, world = 1;
    uint32_t gather_planes = 3;          // planes of the exchange unit: 3 = the quantised framebuffer, 9 = + the parity planes
    uint32_t fb_groups_valid = 3;        // plane groups of d_fb the last scatter wrote (or all three, zeroed, right after init_device_params)
    // last chunk
    uint32_t last_w = 0, last_h = 0, last_offx = 0, last_offy = 0;
    uint32_t tiles_x = 0, tiles_y = 0, n_tiles = 0, tiles_local = 0, tiles_padded = 0;
    // buffers
    uint32_t *d_rng = nullptr;
    float *d_fb = nullptr;          // 9 block-linear planes of n_lanes floats
    float *d_tiles = nullptr;       // compact tile buffer
    size_t tiles_capacity = 0;      // floats
    unsigned long long *d_counters = nullptr;     // [kCounters] statistics + 1 word pixel-queue head behind them
    int n_cu = 256;
    uint32_t waves_per_cu = 0;                         // experiment knob (env SRT_WAVES_PER_CU)
    // step choice of a wave: serve the kind of work (shade / fringe / inner) with the most waiting lanes per unit of cost;
    // weights = 256 / relative cost of the step (inner = 256).
    // Defaults: 70 / 280 when the whole inner tree is LDS resident, 140 / 560 when inner records beyond the cache come from
    // L2 (an inner step then costs about twice as much, so the other two kinds weigh twice as much relative to it);
    // measured plateaus: profiles/r02/knob_sweeps.txt.  0 = not set by the environment.
    uint32_t score_shade = 0, score_fringe = 0;        // env SRT_SCORE_SHADE / SRT_SCORE_FRINGE
    uint32_t debug_lane_limit = 0;                     // test knob (experiments: partial tiles)
    // Test knobs (srt_set_test_knobs; from the environment -- SRT_WIDE_REFS, SRT_LDS_CACHE_MAX, SRT_DEBUG_LANE_LIMIT -- only when
    // SRT_TEST_KNOBS=1, read once here at srt_create): a stray variable in a user's shell cannot change the kernel variant that runs.
    PlanKnobs knobs;
    bool knobs_from_env = false;
    uint32_t split_load_pct = 200;                    // env SRT_SPLIT_LOAD: load factor (%) of the capacity constraint in order_tiles_kernel's split policy (0 = never split)
    uint32_t probe_spp = 2;                            // samples of the cost probe (env SRT_PROBE_SPP, 0 = no ordering)
    // queue order: tile cost moved this % towards 64 x its most expensive pixel (order_tiles_kernel).  -1 = automatic: 100 when the
    // inner tree is partly served by L2 AND the launch has fewer than 6 tiles per persistent wave (a rank's share of an 8-GPU frame:
    // every step of such a chain is an L2 round trip, and a long pixel inside an average tile ends the launch late: cfg 5 at W = 8
    // 2331 -> 2245 ms; with 8 tiles per wave -- cfg 5 at W = 4 -- it already loses: 3665 -> 3812 ms), else 0 (measured worse on LDS-resident
    // trees: cfg 3 at W = 2 187 -> 196 ms, cfg 2 44.8 -> 45.4 ms).
    // env SRT_ORDER_MAX_PCT
    int order_max_pct = -1;
    uint32_t *d_tile_cost = nullptr, *d_tile_order = nullptr;
    size_t tile_sched_capacity = 0;
    uint64_t lanes_allocated = 0;                       // size of d_rng / d_fb in lanes
    float last_probe_ms = 0.f;
    bool count_traversal = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    uint64_t last_paths = 0;
    float *d_rowmajor = nullptr;                        // row-major staging image of srt_read_fb_rowmajor (3 planes)
    OrderProfile order_profile = {};                    // non-zero magic: the next instrumented launch collects the child-order profile
    uint32_t *d_wave_debug = nullptr;                   // instrumented launches: 4 words per wave
    uint32_t wave_debug_waves = 0;
    uint32_t rowmajor_w = 0, rowmajor_h = 0;
};

namespace {

// Optional roctx ranges around the phases of srt_render_chunk (cost probe, queue build, render launch): they show up in
// rocprofv3 --marker-trace timelines next to the kernels.  libroctx64 is looked up once with dlopen; absent library = no ranges.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char *n : {"libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so.4"}) {
            if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) {
                push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
struct RoctxRange {
    static Roctx &api() { static Roctx r; return r; }
    explicit RoctxRange(const char *name) { if (api().push) api().push(name); }
    ~RoctxRange() { if (api().pop) api().pop(); }
};

int fail(srt_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    set_global_error(msg);
    return code;
}
int hip_fail(srt_ctx *ctx, hipError_t e, const char *what) {
    return fail(ctx, SRT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(ctx, expr)                                     \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) return hip_fail(ctx, _e, #expr); \
    } while (0)

template <typename T>
int upload(srt_ctx *ctx, T **dst, const std::vector<float> &src) {
    if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
    HIP_TRY(ctx, hipMalloc((void **)dst, src.size() * sizeof(float)));
    HIP_TRY(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
    return SRT_OK;
}

void fill_params(const srt_ctx *c, RenderParams &p) {
    memset(&p, 0, sizeof(p));
    p.nodes = (const float4 *)c->d_nodes; p.nodes_sw = c->d_nodes_sw; p.fringe = (const float4 *)c->d_fringe; p.tris = (const float4 *)c->d_tris;
    p.mat_sd = (const float2 *)c->d_mat_sd; p.mat_par = (const float4 *)c->d_mat_par;
    p.shade = (const float4 *)c->d_shade; p.cmf = (const float4 *)c->d_cmf;
    p.root_ref = c->root_ref; p.stack_depth = c->stack_depth; p.n_materials = c->n_materials;
    p.n_inner = c->n_inner; p.n_cached = 0;   // n_cached is set by the launcher
    p.n_tris = c->n_tris; p.n_records = c->n_records; p.fringe_stride = c->fringe_stride;
    p.paired = c->paired ? 1u : 0u;
    for (int k = 0; k < 3; k++) {
        p.du[k] = c->cam.pixel_delta_u[k]; p.dv[k] = c->cam.pixel_delta_v[k]; p.p00[k] = c->cam.pixel00_loc[k];
        p.center[k] = c->cam.camera_center[k]; p.disk_u[k] = c->cam.defocus_disk_u[k]; p.disk_v[k] = c->cam.defocus_disk_v[k];
    }
    p.defocus_angle = c->cam.defocus_angle;
    p.tx = c->tx; p.ty = c->ty; p.bx = c->bx; p.by = c->by;
    p.spp = c->spp; p.bounce_limit = c->bounce;
    p.rank = c->rank; p.world = c->world;
    p.rng = c->d_rng; p.n_lanes = c->n_lanes;
    p.tile_out = c->d_tiles; p.counters = c->d_counters;
    p.tile_group_stride = c->tiles_padded * (uint32_t)(kGroupPlanes * kTileLanes);
    p.write_parity = c->gather_planes == (uint32_t)kTilePlanes ? 1u : 0u;
}

}  // namespace

extern "C" {

int srt_create(int device, srt_ctx **out) {
    if (!out) return fail(nullptr, SRT_ERR_INVALID, "srt_create: null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, SRT_ERR_NO_DEVICE, std::string("srt_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") +
                                                    "); this library has no CPU fallback");
    if (device < 0 || device >= n) return fail(nullptr, SRT_ERR_NO_DEVICE, "srt_create: device index out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipSetDevice");
    srt_ctx *c = new srt_ctx();
    c->device = device;
    if (const char *ev = getenv("SRT_WAVES_PER_CU")) c->waves_per_cu = (uint32_t)std::max(0, atoi(ev));
    if (const char *ev = getenv("SRT_PROBE_SPP")) c->probe_spp = (uint32_t)std::max(0, atoi(ev));
    if (const char *ev = getenv("SRT_SCORE_SHADE")) c->score_shade = (uint32_t)std::max(1, atoi(ev));
    if (const char *ev = getenv("SRT_SCORE_FRINGE")) c->score_fringe = (uint32_t)std::max(1, atoi(ev));   // 0 would starve fringe lanes
    if (const char *tk = getenv("SRT_TEST_KNOBS")) if (atoi(tk) == 1) {
        if (const char *ev = getenv("SRT_DEBUG_LANE_LIMIT")) { c->debug_lane_limit = (uint32_t)std::max(0, atoi(ev)); c->knobs_from_env = true; }
        if (const char *ev = getenv("SRT_WIDE_REFS")) { c->knobs.wide_refs = atoi(ev) != 0; c->knobs_from_env = true; }
        if (const char *ev = getenv("SRT_LDS_CACHE_MAX")) { c->knobs.lds_cache_max = std::max(0, atoi(ev)); c->knobs_from_env = true; }
    }
    if (const char *ev = getenv("SRT_SPLIT_LOAD")) c->split_load_pct = (uint32_t)std::max(0, atoi(ev));
    if (const char *ev = getenv("SRT_ORDER_MAX_PCT")) c->order_max_pct = std::min(400, std::max(-1, atoi(ev)));
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
    }
    std::vector<float> rows(96 * 4);
    cmf_rows(rows.data());
    int rc = upload(c, &c->d_cmf, rows);
    if (rc != SRT_OK) { delete c; return rc; }
    if ((e = hipMalloc((void **)&c->d_counters, (kCounters + 1) * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMemset(c->d_counters, 0, (kCounters + 1) * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        int r = hip_fail(nullptr, e, "srt_create");
        srt_destroy(c);
        return r;
    }
    *out = c;
    return SRT_OK;
}

void srt_destroy(srt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    void *bufs[] = {c->d_nodes, c->d_nodes_sw, c->d_fringe, c->d_tris, c->d_mat_sd, c->d_mat_par, c->d_shade, c->d_cmf, c->d_rng, c->d_fb, c->d_tiles, c->d_counters, c->d_tile_cost, c->d_tile_order, c->d_rowmajor, c->d_wave_debug};
    for (void *b : bufs) if (b) (void)hipFree(b);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

int srt_ctx_device(const srt_ctx *ctx) { return ctx ? ctx->device : -1; }
int srt_ctx_cu_count(const srt_ctx *ctx) { return ctx ? ctx->n_cu : -1; }

const char *srt_last_error(const srt_ctx *ctx) { return ctx ? ctx->err.c_str() : global_error(); }

int srt_upload_scene(srt_ctx *c, const srt_scene *s) {
    if (!c || !s) return fail(c, SRT_ERR_INVALID, "srt_upload_scene: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    FlatScene f;
    int rc = flatten_scene(*s, f);
    if (rc != SRT_OK) return fail(c, rc, global_error());
    if (render_lds_bytes(f.stack_depth, 1, 0, f.n_records, c->knobs) > 64 * 1024) return fail(c, SRT_ERR_BVH, "srt_upload_scene: BVH too deep for the LDS traversal stack");
    if ((rc = upload(c, &c->d_nodes, f.nodes)) != SRT_OK) return rc;
    {
        // FRINGE records are 96 B.  Packed, every second one straddles two 128-byte cache lines; when the tree is too large for
        // LDS every visit is an L2 round trip and a record that lies in ONE line halves the lines a FRINGE visit pulls through the
        // CU's small L1: SRT_FRINGE_STRIDE=128 pads such trees' records to a 128-byte stride (measured -1.4 % on cfg 5's scene, kept as a knob).
        LaunchPlan plan;
        render_launch_plan(f.stack_depth, f.n_records, f.n_inner, c->knobs, plan);
        uint32_t stride = plan.all_cached ? 96u : kFringeStrideL2;
        if (const char *ev = getenv("SRT_FRINGE_STRIDE")) stride = (atoi(ev) == 128 && !plan.all_cached) ? 128u : 96u;
        if ((uint64_t)(f.n_records - f.n_inner + 1) * stride >= (1ull << 31)) stride = 96u;
        if (stride == 96u) {
            if ((rc = upload(c, &c->d_fringe, f.fringe)) != SRT_OK) return rc;
        } else {
            const size_t n_fr = f.fringe.size() / 24;
            std::vector<float> padded(n_fr * 32, 0.f);
            for (size_t k = 0; k < n_fr; k++) memcpy(&padded[32 * k], &f.fringe[24 * k], 24 * sizeof(float));
            if ((rc = upload(c, &c->d_fringe, padded)) != SRT_OK) return rc;
        }
        c->fringe_stride = stride;
        // (the pre-swizzled copy of the INNER records: only trees whose INNER visits read memory need it)
        if (c->d_nodes_sw) { (void)hipFree(c->d_nodes_sw); c->d_nodes_sw = nullptr; }
        if (!plan.all_cached && (rc = upload(c, &c->d_nodes_sw, f.nodes_sw)) != SRT_OK) return rc;
    }
    if ((rc = upload(c, &c->d_tris, f.tris)) != SRT_OK) return rc;
    if ((rc = upload(c, &c->d_mat_sd, f.mat_sd)) != SRT_OK) return rc;
    if ((rc = upload(c, &c->d_mat_par, f.mat_par)) != SRT_OK) return rc;
    if ((rc = upload(c, &c->d_shade, f.shade)) != SRT_OK) return rc;
    c->root_ref = f.root_ref; c->stack_depth = f.stack_depth; c->n_materials = (uint32_t)s->mats.size();
    c->n_inner = f.n_inner; c->n_records = f.n_records; c->n_tris = (uint32_t)s->raw.size();
    c->paired = tree_is_paired(*s);
    c->scene_ready = true;
    return SRT_OK;
}

int srt_set_camera(srt_ctx *c, const srt_camera_data *cam) {
    if (!c || !cam) return fail(c, SRT_ERR_INVALID, "srt_set_camera: null argument");
    c->cam = *cam;
    c->camera_ready = true;
    return SRT_OK;
}

int srt_launch_plan(const srt_ctx *c, int *waves_per_cu, int *n_cached, int *all_cached, int *narrow_refs) {
    if (!c || !c->scene_ready) return fail(nullptr, SRT_ERR_INVALID, "srt_launch_plan: no scene uploaded");
    LaunchPlan plan;
    render_launch_plan(c->stack_depth, c->n_records, c->n_inner, c->knobs, plan);
    if (waves_per_cu) *waves_per_cu = plan.waves_per_cu;
    if (n_cached) *n_cached = plan.n_cached;
    if (all_cached) *all_cached = plan.all_cached ? 1 : 0;
    if (narrow_refs) *narrow_refs = render_narrow_refs(c->n_records, c->knobs) ? 1 : 0;
    return SRT_OK;
}

int srt_launch_paired(const srt_ctx *c, int *paired) {
    if (!c || !c->scene_ready || !paired) return fail(nullptr, SRT_ERR_INVALID, "srt_launch_paired: no scene uploaded / null argument");
    LaunchPlan plan;
    render_launch_plan(c->stack_depth, c->n_records, c->n_inner, c->knobs, plan);
    *paired = render_paired_variant(c->paired, render_narrow_refs(c->n_records, c->knobs), plan.all_cached) ? 1 : 0;
    return SRT_OK;
}

int srt_launch_lds_bytes(const srt_ctx *c, size_t *bytes) {
    if (!c || !c->scene_ready || !bytes) return fail(nullptr, SRT_ERR_INVALID, "srt_launch_lds_bytes: no scene uploaded / null argument");
    LaunchPlan plan;
    render_launch_plan(c->stack_depth, c->n_records, c->n_inner, c->knobs, plan);
    *bytes = render_lds_bytes(c->stack_depth, plan.waves_per_block, plan.n_cached, c->n_records, c->knobs);
    return SRT_OK;
}

// Test knobs of a context: wide_refs != 0 sends small trees through the 32-bit-reference kernel variants, lds_cache_max >= 0 caps the
// inner records kept in LDS (0 = every inner record from L2), lane_limit > 0 renders only the first lane_limit pixels of every tile.
// -1 / 0 / 0 restores the defaults.  A scene uploaded before the call must be uploaded again (the FRINGE stride follows the plan).
int srt_set_test_knobs(srt_ctx *c, int wide_refs, int lds_cache_max, uint32_t lane_limit) {
    if (!c || lds_cache_max < -1 || lane_limit > 64) return fail(c, SRT_ERR_INVALID, "srt_set_test_knobs: bad argument");
    c->knobs.wide_refs = wide_refs != 0; c->knobs.lds_cache_max = lds_cache_max; c->debug_lane_limit = lane_limit;
    c->scene_ready = false;      // the upload's FRINGE stride and the plan must be made with the same knobs
    return SRT_OK;
}
int srt_get_test_knobs(const srt_ctx *c, int *wide_refs, int *lds_cache_max, uint32_t *lane_limit, int *from_env) {
    if (!c) return fail(nullptr, SRT_ERR_INVALID, "srt_get_test_knobs: null ctx");
    if (wide_refs) *wide_refs = c->knobs.wide_refs ? 1 : 0;
    if (lds_cache_max) *lds_cache_max = c->knobs.lds_cache_max;
    if (lane_limit) *lane_limit = c->debug_lane_limit;
    if (from_env) *from_env = c->knobs_from_env ? 1 : 0;
    return SRT_OK;
}

// srt_init_device_params without the final device-wide wait: a communicator that drives several GPUs from one process enqueues
// the re-seeding on all of them before it waits for any (srt_comm_init_device_params)
int srt_internal_init_device_params(srt_ctx *c, uint32_t tx, uint32_t ty, uint32_t bx, uint32_t by, uint32_t chunk_w, uint32_t chunk_h,
                                    uint32_t spp, uint32_t bounce_limit, uint64_t seed, int wait) {
    if (!c) return fail(c, SRT_ERR_INVALID, "srt_init_device_params: null ctx");
    if (tx == 0 || ty == 0 || bx == 0 || by == 0 || chunk_w == 0 || chunk_h == 0)
        return fail(c, SRT_ERR_INVALID, "srt_init_device_params: zero dimension");
    const uint64_t lanes = (uint64_t)tx * ty * bx * by;
    if (lanes > 0x7fffffffull) return fail(c, SRT_ERR_INVALID, "srt_init_device_params: grid too large");
    HIP_TRY(c, hipSetDevice(c->device));
    c->tx = tx; c->ty = ty; c->bx = bx; c->by = by; c->chunk_w = chunk_w; c->chunk_h = chunk_h;
    c->spp = (uint16_t)spp; c->bounce = (uint16_t)bounce_limit;     // short_uint, rendering.cu:154 (Q17)
    c->seed = seed; c->n_lanes = (uint32_t)lanes;
    if (lanes != c->lanes_allocated) {   // a new frame of the same grid re-seeds in place
        if (c->d_rng) { (void)hipFree(c->d_rng); c->d_rng = nullptr; }
        if (c->d_fb) { (void)hipFree(c->d_fb); c->d_fb = nullptr; }
        c->lanes_allocated = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_rng, 6 * lanes * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void **)&c->d_fb, kTilePlanes * lanes * sizeof(float)));
        c->lanes_allocated = lanes;
    }
    HIP_TRY(c, hipMemset(c->d_fb, 0, kTilePlanes * lanes * sizeof(float)));
    HIP_TRY(c, launch_init_rng(c->d_rng, c->n_lanes, seed, nullptr));   // init_random_states, rendering.cu:330
    if (wait) HIP_TRY(c, hipDeviceSynchronize());
    c->fb_groups_valid = (uint32_t)kTileGroups;      // all nine planes are zero
    c->params_ready = true;
    return SRT_OK;
}

int srt_init_device_params(srt_ctx *c, uint32_t tx, uint32_t ty, uint32_t bx, uint32_t by, uint32_t chunk_w, uint32_t chunk_h,
                           uint32_t spp, uint32_t bounce_limit, uint64_t seed) {
    return srt_internal_init_device_params(c, tx, ty, bx, by, chunk_w, chunk_h, spp, bounce_limit, seed, 1);
}

int srt_set_partition(srt_ctx *c, uint32_t rank, uint32_t world) {
    if (!c || world == 0 || rank >= world) return fail(c, SRT_ERR_INVALID, "srt_set_partition: need rank < world");
    c->rank = rank; c->world = world;
    return SRT_OK;
}

uint32_t srt_internal_gather_planes(const srt_ctx *c) { return c ? c->gather_planes : 0u; }

int srt_set_gather_planes(srt_ctx *c, uint32_t planes) {
    if (!c || (planes != 3 && planes != 9)) return fail(c, SRT_ERR_INVALID, "srt_set_gather_planes: planes must be 3 or 9");
    c->gather_planes = planes;
    return SRT_OK;
}

int srt_render_chunk(srt_ctx *c, uint32_t width, uint32_t height, uint32_t offx, uint32_t offy, void *stream) {
    if (!c) return fail(c, SRT_ERR_INVALID, "srt_render_chunk: null ctx");
    // reference: "Device parameters were not initialized, render aborted" (rendering.cu:247-250)
    if (!c->scene_ready || !c->camera_ready || !c->params_ready)
        return fail(c, SRT_ERR_INVALID, "srt_render_chunk: scene, camera and device parameters must be set first");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    width = (uint16_t)width; height = (uint16_t)height; offx = (uint16_t)offx; offy = (uint16_t)offy;   // rendering.cu:245 (Q17)
    c->last_w = width; c->last_h = height; c->last_offx = offx; c->last_offy = offy;
    // Tiles cover every pixel the reference grid can address, whatever the size of THIS chunk: the tile number of a lane
    // idx -- and with it the rank that owns the lane's persistent RNG stream (Q13) -- must not move when a ragged edge
    // chunk is narrower than the one before.  Tiles (partly) outside the chunk just skip those pixels (rendering.cu:205).
    const uint32_t cover_w = c->tx * c->bx, cover_h = c->ty * c->by;
    c->tiles_x = (cover_w + 7) / 8; c->tiles_y = (cover_h + 7) / 8;
    c->n_tiles = c->tiles_x * c->tiles_y;
    c->tiles_padded = (c->n_tiles + c->world - 1) / c->world;
    c->tiles_local = c->n_tiles > c->rank ? (c->n_tiles - c->rank + c->world - 1) / c->world : 0;
    const size_t need = (size_t)std::max<uint32_t>(c->tiles_padded, 1) * kTilePlanes * kTileLanes;
    if (need > c->tiles_capacity) {
        if (c->d_tiles) { (void)hipFree(c->d_tiles); c->d_tiles = nullptr; }
        HIP_TRY(c, hipMalloc((void **)&c->d_tiles, need * sizeof(float)));
        c->tiles_capacity = need;
    }
    // (only the plane groups this launch writes: group 0, or all three when the parity planes were asked for)
    HIP_TRY(c, hipMemsetAsync(c->d_tiles, 0, (size_t)std::max<uint32_t>(c->tiles_padded, 1) * c->gather_planes * kTileLanes * sizeof(float), st));
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, (kCounters + 1) * sizeof(unsigned long long), st));
    RenderParams p;
    fill_params(c, p);
    p.width = width; p.height = height; p.offx = offx; p.offy = offy;
    p.tiles_x = c->tiles_x; p.tiles_y = c->tiles_y; p.n_tiles = c->n_tiles;
    p.tiles_local = c->tiles_local;
    p.pixel_counter = (uint32_t *)(c->d_counters + kCounters);
    p.waves_per_cu_override = c->waves_per_cu;
    LaunchPlan plan;
    render_launch_plan(c->stack_depth, c->n_records, c->n_inner, c->knobs, plan);
    {
        const bool all_cached = plan.all_cached;
        // (inner records that come from L2 make an INNER visit ~2x as expensive, so shading and FRINGE visits weigh more:
        // plateau 280-400 / 560-1100 on cfg 5's scene, 60-85 / 280-340 on cfg 2 / 3 / 4, profiles/r02/knob_sweeps.txt)
        p.score_shade = c->score_shade ? c->score_shade : (all_cached ? 70u : kScoreShadeL2);
        // (the PAIRED variant's FRINGE visit is a fifth cheaper: plateau 340-480 on cfg 3, profiles/r05/experiments/weights_paired.txt)
        const bool paired = render_paired_variant(c->paired, render_narrow_refs(c->n_records, c->knobs), all_cached);
        p.score_fringe = c->score_fringe ? c->score_fringe : (all_cached ? (paired ? 400u : 280u) : kScoreFringeL2);
    }
    // ---- cost-ordered pixel queue --------------------------------------------------------------------------------
    // A pixel is one sequential RNG stream, so the launch cannot finish before its most expensive pixel does.  A short
    // probe (probe_spp samples per pixel from a copy of the RNG state, nothing written) measures the traversal cost of
    // every tile; order_tiles_kernel then builds the queue on the device: tiles in descending cost order
    // (longest-processing-time-first), the most expensive ones split over several waves when the launch is chain-bound.
    p.tile_order = nullptr; p.tile_cost = nullptr; p.queue_rows = nullptr; p.queue_rows_bound = c->tiles_local;
    p.debug_lane_limit = c->debug_lane_limit;
    const bool ordered = c->probe_spp > 0 && c->spp > 4 * c->probe_spp && c->tiles_local > 1 && c->tiles_local <= 0x3fffffu;   // 22-bit tile field of a queue row
    if (ordered) {
        if (c->tiles_local > c->tile_sched_capacity) {
            if (c->d_tile_cost) { (void)hipFree(c->d_tile_cost); c->d_tile_cost = nullptr; }
            if (c->d_tile_order) { (void)hipFree(c->d_tile_order); c->d_tile_order = nullptr; }
            HIP_TRY(c, hipMalloc((void **)&c->d_tile_cost, 2 * (size_t)c->tiles_local * sizeof(uint32_t)));      // cost | most expensive pixel
            // [rows: up to 64 per tile][sorted tile ids][queue_info]
            HIP_TRY(c, hipMalloc((void **)&c->d_tile_order, ((size_t)c->tiles_local * 65 + 4) * sizeof(uint32_t)));
            c->tile_sched_capacity = c->tiles_local;
        }
        uint32_t *rows = c->d_tile_order, *sorted = rows + (size_t)c->tile_sched_capacity * 64, *queue_info = sorted + c->tile_sched_capacity;
        HIP_TRY(c, hipMemsetAsync(c->d_tile_cost, 0, 2 * (size_t)c->tiles_local * sizeof(uint32_t), st));
        RenderParams pp = p;
        pp.spp = c->probe_spp; pp.tile_cost = c->d_tile_cost;
        RoctxRange range_probe("srt cost probe + pixel queue");
        HIP_TRY(c, launch_render(pp, c->knobs, (uint32_t)c->n_cu, 2, st));
        const uint32_t split_pct = c->split_load_pct;
        const uint32_t n_waves_plan = (uint32_t)c->n_cu * (uint32_t)plan.waves_per_cu;
        const uint32_t order_pct = c->order_max_pct >= 0 ? (uint32_t)c->order_max_pct : ((!plan.all_cached && (uint64_t)c->tiles_local < 6ull * n_waves_plan) ? 100u : 0u);
        HIP_TRY(c, launch_order_tiles(c->d_tile_cost, sorted, rows, c->tiles_local, (uint32_t)c->n_cu * (uint32_t)plan.waves_per_cu, split_pct, queue_info, order_pct, st));   // device-side, no host sync
        HIP_TRY(c, hipMemsetAsync(c->d_counters + kCounters, 0, sizeof(unsigned long long), st));   // rewind the queue head
        p.tile_order = rows;
        p.queue_rows = queue_info;
        p.prio_cost = c->d_tile_cost;      // wave priorities of the render launch (render_kernel, LDS-resident trees)
        if (split_pct) p.queue_rows_bound = (uint32_t)std::min<uint64_t>((uint64_t)c->tiles_local * 64, 0x7fffffffull);
    }
    if (c->count_traversal) {
        const uint32_t n_waves = (uint32_t)c->n_cu * (uint32_t)plan.waves_per_cu;
        // layout of the debug buffer: [OrderProfile header][4 words per wave]: the header sits at a FIXED place, so the kernel finds it
        // whatever number of waves the launcher ends up starting
        if (n_waves > c->wave_debug_waves) {
            if (c->d_wave_debug) { (void)hipFree(c->d_wave_debug); c->d_wave_debug = nullptr; c->wave_debug_waves = 0; }
            HIP_TRY(c, hipMalloc((void **)&c->d_wave_debug, sizeof(OrderProfile) + (size_t)n_waves * 4 * sizeof(uint32_t)));
            c->wave_debug_waves = n_waves;
        }
        HIP_TRY(c, hipMemsetAsync(c->d_wave_debug, 0, sizeof(OrderProfile) + (size_t)c->wave_debug_waves * 4 * sizeof(uint32_t), st));
        p.wave_debug = c->d_wave_debug;
        if (c->order_profile.magic == kOrderProfileMagic)
            HIP_TRY(c, hipMemcpyAsync(c->d_wave_debug, &c->order_profile, sizeof(OrderProfile), hipMemcpyHostToDevice, st));
    }
    RoctxRange range_render("srt render_kernel");
    HIP_TRY(c, hipEventRecord(c->ev0, st));     // ev0..ev1 bracket the render kernel alone (roofline.achieved)
    uint32_t waves_launched = 0;
    HIP_TRY(c, launch_render(p, c->knobs, (uint32_t)c->n_cu, c->count_traversal ? 1 : 0, st, &waves_launched));
    if (c->count_traversal && waves_launched > c->wave_debug_waves)
        return fail(c, SRT_ERR_HIP, "srt_render_chunk: the launch started more waves than the debug buffer holds (launch plan and launcher disagree)");
    HIP_TRY(c, hipEventRecord(c->ev1, st));
    c->timed = true;
    c->last_paths = 0;   // filled by srt_get_stats from the tile ownership
    return SRT_OK;
}

int srt_synchronize(srt_ctx *c) {
    if (!c) return fail(c, SRT_ERR_INVALID, "srt_synchronize: null ctx");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    return SRT_OK;
}

int srt_tile_buffer(srt_ctx *c, void **dev_ptr, size_t *n_floats, uint32_t *tiles_local, uint32_t *tiles_padded) {
    if (!c || !c->d_tiles) return fail(c, SRT_ERR_INVALID, "srt_tile_buffer: nothing rendered yet");
    if (dev_ptr) *dev_ptr = c->d_tiles;
    if (n_floats) *n_floats = (size_t)c->tiles_padded * c->gather_planes * kTileLanes;      // the exchange unit: the first 1 or 3 plane groups
    if (tiles_local) *tiles_local = c->tiles_local;
    if (tiles_padded) *tiles_padded = c->tiles_padded;
    return SRT_OK;
}

int srt_copy_tile_buffer(srt_ctx *c, void *dst_dev, void *stream) {
    if (!c || !c->d_tiles || !dst_dev) return fail(c, SRT_ERR_INVALID, "srt_copy_tile_buffer: nothing rendered yet / null destination");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(dst_dev, c->d_tiles, (size_t)c->tiles_padded * c->gather_planes * kTileLanes * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SRT_OK;
}

int srt_scatter_tiles(srt_ctx *c, const void *dev_gathered, void *stream) {
    if (!c || !c->d_fb || !c->d_tiles) return fail(c, SRT_ERR_INVALID, "srt_scatter_tiles: nothing rendered yet");
    uint32_t groups = c->gather_planes / (uint32_t)kGroupPlanes;
    if (!dev_gathered) {
        if (c->world != 1) return fail(c, SRT_ERR_INVALID, "srt_scatter_tiles: a gathered buffer is required when world > 1");
        dev_gathered = c->d_tiles;      // the context's own tile buffer: group 0, and the parity groups when they were asked for
    }
    HIP_TRY(c, hipSetDevice(c->device));
    ScatterParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.gathered = (const float *)dev_gathered;
    sp.groups = groups;
    for (int p = 0; p < kTilePlanes; p++) sp.fb[p] = c->d_fb + (size_t)p * c->n_lanes;
    sp.width = c->last_w; sp.height = c->last_h;
    sp.tx = c->tx; sp.ty = c->ty; sp.bx = c->bx; sp.by = c->by;
    sp.tiles_x = c->tiles_x; sp.n_tiles = c->n_tiles; sp.world = c->world; sp.tiles_padded = c->tiles_padded;
    HIP_TRY(c, launch_scatter(sp, (hipStream_t)stream));
    c->fb_groups_valid = groups;      // with a 3-plane exchange unit the parity planes of d_fb are NOT those of this frame
    return SRT_OK;
}

int srt_dev_fb(srt_ctx *c, void **r, void **g, void **b, size_t *n_floats) {
    if (!c || !c->d_fb) return fail(c, SRT_ERR_INVALID, "srt_dev_fb: device parameters not initialised");
    if (r) *r = c->d_fb;
    if (g) *g = c->d_fb + (size_t)c->n_lanes;
    if (b) *b = c->d_fb + 2 * (size_t)c->n_lanes;
    if (n_floats) *n_floats = c->n_lanes;
    return SRT_OK;
}

static int read_planes(srt_ctx *c, int first_plane, float *p0, float *p1, float *p2) {
    if (!c || !c->d_fb) return fail(c, SRT_ERR_INVALID, "read: device parameters not initialised");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    float *dst[3] = {p0, p1, p2};
    for (int k = 0; k < 3; k++)
        if (dst[k]) HIP_TRY(c, hipMemcpy(dst[k], c->d_fb + (size_t)(first_plane + k) * c->n_lanes, (size_t)c->n_lanes * sizeof(float), hipMemcpyDeviceToHost));
    return SRT_OK;
}

int srt_read_fb(srt_ctx *c, float *r, float *g, float *b) { return read_planes(c, 0, r, g, b); }

int srt_read_fb_aux(srt_ctx *c, int which, float *p0, float *p1, float *p2) {
    if (which != 1 && which != 2) return fail(c, SRT_ERR_INVALID, "srt_read_fb_aux: which must be 1 (sRGB) or 2 (XYZ)");
    if (c && (uint32_t)which >= c->fb_groups_valid)
        return fail(c, SRT_ERR_UNSUPPORTED, "srt_read_fb_aux: the parity planes of the last frame were not gathered (the exchange unit was the 3 quantised "
                                            "planes; srt_set_gather_planes / srt_comm_set_gather_planes(.., 9) before rendering moves all nine)");
    return read_planes(c, 3 * which, p0, p1, p2);
}

int srt_read_fb_rowmajor(srt_ctx *c, float *r, float *g, float *b, uint32_t image_width, uint32_t image_height) {
    if (!c || !c->d_fb || !r || !g || !b || image_width == 0 || image_height == 0) return fail(c, SRT_ERR_INVALID, "srt_read_fb_rowmajor: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)image_width * image_height;
    // context-owned row-major staging image: the un-swizzle writes the chunk's pixels into it on the device and only the
    // chunk's rectangle travels to the caller's planes (update_fb touches nothing else either, render_manager.cuh:68-142)
    if (c->rowmajor_w != image_width || c->rowmajor_h != image_height || !c->d_rowmajor) {
        if (c->d_rowmajor) { (void)hipFree(c->d_rowmajor); c->d_rowmajor = nullptr; }
        HIP_TRY(c, hipMalloc((void **)&c->d_rowmajor, 3 * n * sizeof(float)));
        HIP_TRY(c, hipMemset(c->d_rowmajor, 0, 3 * n * sizeof(float)));
        c->rowmajor_w = image_width; c->rowmajor_h = image_height;
    }
    const float *src[3] = {c->d_fb, c->d_fb + (size_t)c->n_lanes, c->d_fb + 2 * (size_t)c->n_lanes};
    float *dst[3] = {c->d_rowmajor, c->d_rowmajor + n, c->d_rowmajor + 2 * n};
    HIP_TRY(c, launch_unswizzle(src, dst, c->tx, c->ty, c->bx, c->by, c->last_w, c->last_h, c->last_offx, c->last_offy, image_width, image_height, nullptr));
    if (c->last_offx < image_width && c->last_offy < image_height) {
        const uint32_t w = std::min<uint32_t>(std::min<uint32_t>(c->last_w, c->tx * c->bx), image_width - c->last_offx);
        const uint32_t h = std::min<uint32_t>(std::min<uint32_t>(c->last_h, c->ty * c->by), image_height - c->last_offy);
        const size_t first = (size_t)c->last_offy * image_width + c->last_offx, pitch = (size_t)image_width * sizeof(float);
        float *host[3] = {r, g, b};
        for (int k = 0; k < 3 && w && h; k++)
            HIP_TRY(c, hipMemcpy2D(host[k] + first, pitch, dst[k] + first, pitch, (size_t)w * sizeof(float), h, hipMemcpyDeviceToHost));
    }
    HIP_TRY(c, hipDeviceSynchronize());
    return SRT_OK;
}

int srt_get_tile_costs(srt_ctx *c, uint32_t *out, size_t n) {
    // n = tiles_local: the probe's cost per local tile; n = 2 * tiles_local: followed by the cost of each tile's most expensive pixel
    if (!c || !out || !c->d_tile_cost || (n > c->tiles_local && n != 2 * (size_t)c->tiles_local)) return fail(c, SRT_ERR_INVALID, "srt_get_tile_costs: no probe has run / bad size");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(out, c->d_tile_cost, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SRT_OK;
}

int srt_get_stats(srt_ctx *c, srt_stats *out) {
    if (!c || !out) return fail(c, SRT_ERR_INVALID, "srt_get_stats: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    unsigned long long h[kCounters];
    HIP_TRY(c, hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    if (h[23] != 0)      // instrumented launches: render_kernel S3's invariant (a wave's first fetch is made by all 64 lanes)
        return fail(c, SRT_ERR_HIP, "srt_get_stats: a wave made a partial first fetch from the pixel queue: its assigned first row was not rendered");
    memset(out, 0, sizeof(*out));
    out->rays = h[0]; out->node_visits = h[1]; out->tri_tests = h[2]; out->box_tests = h[3];
    for (int k = 0; k < 9; k++) out->util[k] = h[4 + k];
    out->reserved[0] = h[13]; out->reserved[1] = h[14];
    for (int k = 0; k < 4; k++) out->shade[k] = h[15 + k];
    for (int k = 0; k < 4; k++) out->waves[k] = h[19 + k];   // instrumented: waves, sum / max of their life times, drain time
    out->hits = h[24];
    // paths = spp * pixels owned by this rank
    uint64_t pixels = 0;
    for (uint32_t t = c->rank; t < c->n_tiles; t += c->world) {
        const uint32_t tx0 = (t % c->tiles_x) * 8, ty0 = (t / c->tiles_x) * 8;
        const uint32_t lim_w = std::min<uint32_t>(c->last_w, c->tx * c->bx), lim_h = std::min<uint32_t>(c->last_h, c->ty * c->by);
        const uint32_t w = tx0 < lim_w ? std::min<uint32_t>(8, lim_w - tx0) : 0, hgt = ty0 < lim_h ? std::min<uint32_t>(8, lim_h - ty0) : 0;
        pixels += (uint64_t)w * hgt;
    }
    out->paths = pixels * c->spp;
    return SRT_OK;
}

int srt_get_wave_debug(srt_ctx *c, uint32_t *out, size_t n_waves) {
    if (!c || !out || !c->d_wave_debug || n_waves > c->wave_debug_waves) return fail(c, SRT_ERR_INVALID, "srt_get_wave_debug: no instrumented launch yet / bad size");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(out, reinterpret_cast<const char *>(c->d_wave_debug) + sizeof(OrderProfile), n_waves * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SRT_OK;
}

// Child order from a profile of the real rays.  The traversal is the reference's fixed left-first walk (bvh.cu:154-160), so which
// child of a node is visited first is a property of the TREE -- and for every ray whose closest hit lies under one child while the
// other child's box lies on its path beyond that hit, visiting the hit's side first lets closest_so_far prune the other subtree.
// The builder's rule (nearer child to the camera first) is right for camera rays; this call measures instead: one instrumented
// probe frame of the context's camera (width x height, spp, bounce_limit) in which every finished closest-hit query walks from
// its triangle's leaf to the root and notes, at every ancestor whose OTHER child's box the ray meets only beyond the hit, under
// which child the hit lay.  Children are swapped wherever the right one won more often (at least `min_samples` such rays at the
// node; elsewhere the existing order stays).  A second probe frame then checks the work counters (node records + 2 x triangle
// tests of the same frame, deterministic): if the new order is not cheaper the swaps are undone (cfg 5's mesh: no gain -> kept as
// built).  The scene is left (re-)ordered and uploaded; topology, boxes and depth do not change, and like any change of the tree
// it can only alter a result where two triangles tie exactly in t (Q11).  n_swapped (optional) = nodes changed (0 after an undo).
int srt_order_children_by_profile(srt_ctx *c, srt_scene *s, uint32_t width, uint32_t height, uint32_t spp, uint32_t bounce_limit,
                                  uint32_t min_samples, uint32_t *n_swapped) {
    if (n_swapped) *n_swapped = 0;
    if (!c || !s || !s->bvh_valid) return fail(c, SRT_ERR_INVALID, "srt_order_children_by_profile: null argument / BVH not built");
    if (!c->camera_ready) return fail(c, SRT_ERR_INVALID, "srt_order_children_by_profile: set the camera first (srt_set_camera)");
    if (width == 0 || height == 0 || spp == 0) return fail(c, SRT_ERR_INVALID, "srt_order_children_by_profile: empty probe frame");
    const size_t n_nodes = s->nodes.size(), n_tris = s->raw.size();
    int rc = srt_upload_scene(c, s);
    if (rc != SRT_OK || n_nodes < 3) return rc;
    std::vector<int32_t> leaf(n_tris, 0), up(n_nodes, -1);
    std::vector<float> sibbox(6 * n_nodes, 0.f);
    for (size_t k = 0; k < n_nodes; k++) {
        const BvhNode &nd = s->nodes[k];
        if (nd.prim >= 0) { leaf[(size_t)nd.prim] = (int32_t)k; continue; }
        up[(size_t)nd.left] = (int32_t)(2 * k); up[(size_t)nd.right] = (int32_t)(2 * k + 1);
        memcpy(&sibbox[6 * (size_t)nd.left], s->nodes[(size_t)nd.right].box, 6 * sizeof(float));
        memcpy(&sibbox[6 * (size_t)nd.right], s->nodes[(size_t)nd.left].box, 6 * sizeof(float));
    }
    int32_t *d_leaf = nullptr, *d_up = nullptr; float *d_sib = nullptr; uint32_t *d_cnt = nullptr;
    auto release = [&]() { (void)hipFree(d_leaf); (void)hipFree(d_up); (void)hipFree(d_sib); (void)hipFree(d_cnt); c->order_profile = OrderProfile{}; };
    HIP_TRY(c, hipSetDevice(c->device));
    hipError_t e = hipMalloc((void **)&d_leaf, std::max<size_t>(n_tris, 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&d_up, n_nodes * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&d_sib, 6 * n_nodes * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&d_cnt, 2 * n_nodes * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(d_leaf, leaf.data(), n_tris * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_up, up.data(), n_nodes * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_sib, sibbox.data(), 6 * n_nodes * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_cnt, 0, 2 * n_nodes * sizeof(uint32_t));
    if (e != hipSuccess) { release(); return hip_fail(c, e, "srt_order_children_by_profile: device buffers"); }
    // instrumented probe frames on this context alone (its partition and counter setting are restored afterwards)
    const bool counting = c->count_traversal;
    const uint32_t rank = c->rank, world = c->world;
    const uint32_t tx = 28, ty = 16, bx = width / tx + 1, by = height / ty + 1;      // the reference's grid (render_manager.cu:84-94)
    auto probe_frame = [&](bool collect, unsigned long long &work) -> int {
        c->order_profile = OrderProfile{};
        if (collect) {
            c->order_profile.magic = kOrderProfileMagic; c->order_profile.leaf = d_leaf; c->order_profile.up = d_up;
            c->order_profile.sibbox = d_sib; c->order_profile.cnt = d_cnt; c->order_profile.n_nodes = n_nodes;
        }
        c->count_traversal = true; c->rank = 0; c->world = 1;
        int r = srt_init_device_params(c, tx, ty, bx, by, width, height, spp, bounce_limit, 1984);
        if (r == SRT_OK) r = srt_render_chunk(c, width, height, 0, 0, nullptr);
        if (r == SRT_OK) r = srt_synchronize(c);
        c->count_traversal = counting; c->rank = rank; c->world = world;
        c->order_profile = OrderProfile{};
        c->params_ready = false;      // (the probe's RNG state and grid are not the caller's: srt_init_device_params comes next)
        if (r != SRT_OK) return r;
        unsigned long long h[kCounters];
        const hipError_t he = hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost);
        if (he != hipSuccess) return hip_fail(c, he, "srt_order_children_by_profile: counters");
        work = h[1] + 2ull * h[2];      // node records visited + 2 x triangle tests
        return SRT_OK;
    };
    unsigned long long work_before = 0, work_after = 0;
    rc = probe_frame(true, work_before);
    std::vector<uint32_t> cnt(2 * n_nodes, 0u);
    if (rc == SRT_OK) { e = hipMemcpy(cnt.data(), d_cnt, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost); if (e != hipSuccess) rc = hip_fail(c, e, "srt_order_children_by_profile: read back"); }
    if (rc != SRT_OK) { release(); return rc; }
    {
        // a collecting launch that recorded nothing (header not seen by the kernel, or a probe frame without a single hit) must not read
        // as "the order was already the cheapest"
        unsigned long long samples = 0;
        for (uint32_t v : cnt) samples += v;
        if (samples == 0) { release(); return fail(c, SRT_ERR_INVALID, "srt_order_children_by_profile: the probe frame recorded no samples (no closest hit with the sibling's box beyond it)"); }
    }
    const uint32_t *won = cnt.data();      // [node * 2 + side]: hits under that child with the sibling's box beyond the hit
    std::vector<uint32_t> swapped_nodes;
    for (size_t k = 0; k < n_nodes; k++) {
        BvhNode &nd = s->nodes[k];
        if (nd.prim >= 0) continue;
        const uint32_t l = won[2 * k], r = won[2 * k + 1];
        if (l + r >= std::max<uint32_t>(min_samples, 1u) && r > l) { std::swap(nd.left, nd.right); swapped_nodes.push_back((uint32_t)k); }
    }
    if (!swapped_nodes.empty()) {
        rc = srt_upload_scene(c, s);
        if (rc == SRT_OK) rc = probe_frame(false, work_after);
        if (rc == SRT_OK && work_after >= work_before) {      // not cheaper on the very frame it was derived from: undo
            for (uint32_t k : swapped_nodes) std::swap(s->nodes[k].left, s->nodes[k].right);
            swapped_nodes.clear();
            rc = srt_upload_scene(c, s);
        }
    }
    release();
    if (rc == SRT_OK && n_swapped) *n_swapped = (uint32_t)swapped_nodes.size();
    return rc;
}

int srt_set_count_traversal(srt_ctx *c, int on) {
    if (!c) return fail(c, SRT_ERR_INVALID, "srt_set_count_traversal: null ctx");
    c->count_traversal = on != 0;
    return SRT_OK;
}

int srt_last_kernel_ms(srt_ctx *c, float *ms) {
    if (!c || !ms || !c->timed) return fail(c, SRT_ERR_INVALID, "srt_last_kernel_ms: nothing rendered yet");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return SRT_OK;
}

int srt_trace_rays(srt_ctx *c, const float *rays, size_t n, float *out) {
    if (!c || !c->scene_ready || (!rays && n) || (!out && n)) return fail(c, SRT_ERR_INVALID, "srt_trace_rays: scene not uploaded / bad argument");
    if (n == 0) return SRT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    float *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_in, 6 * n * sizeof(float)));
    hipError_t e = hipMalloc((void **)&d_out, 4 * n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_in, rays, 6 * n * sizeof(float), hipMemcpyHostToDevice);
    RenderParams p;
    fill_params(c, p);
    if (e == hipSuccess) e = launch_trace(p, d_in, n, d_out, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d_out, 4 * n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return hip_fail(c, e, "srt_trace_rays");
    return SRT_OK;
}

int srt_device_op_sweep(srt_ctx *c, int which, const float *a, const float *b, size_t n, float *out) {
    if (!c || !a || !b || !out) return fail(c, SRT_ERR_INVALID, "srt_device_op_sweep: null argument");
    if (n == 0) return SRT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    float *d = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d, 3 * n * sizeof(float)));
    hipError_t e = hipMemcpy(d, a, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_op_sweep(which, d, d + n, n, d + 2 * n, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(c, e, "srt_device_op_sweep");
    return SRT_OK;
}

int srt_calibrate(srt_ctx *c, int kind, uint32_t waves_per_simd, uint32_t iters, srt_calibration *out) {
    if (!c || !out || kind < 0 || kind >= calib_kinds() || waves_per_simd < 1 || waves_per_simd > 4 || iters == 0)
        return fail(c, SRT_ERR_INVALID, "srt_calibrate: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t threads = waves_per_simd * 256u, n_blocks = (uint32_t)c->n_cu, n_waves = n_blocks * threads / 64u;
    unsigned long long *d_cyc = nullptr;
    float *d_sink = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_cyc, n_waves * sizeof(unsigned long long)));
    hipError_t e = hipMalloc((void **)&d_sink, 1024 * sizeof(float));
    // kinds 11+: a 16 MB table of 64-byte records (the size of the 100k-triangle mesh's tree: served by L2 / MALL)
    const uint32_t n_table = 1u << 18;
    float4 *d_table = nullptr;
    if (e == hipSuccess && kind >= 11) { e = hipMalloc((void **)&d_table, (size_t)n_table * 64); if (e == hipSuccess) e = hipMemset(d_table, 0, (size_t)n_table * 64); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = launch_calib(kind, n_blocks, threads, iters / 8u + 1u, d_sink, d_cyc, d_table, n_table, nullptr);   // warm-up (clocks, code)
    if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
    if (e == hipSuccess) e = launch_calib(kind, n_blocks, threads, iters, d_sink, d_cyc, d_table, n_table, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(n_waves);
    if (e == hipSuccess) e = hipMemcpy(h.data(), d_cyc, n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_cyc);
    if (d_sink) (void)hipFree(d_sink);
    if (d_table) (void)hipFree(d_table);
    if (e != hipSuccess) return hip_fail(c, e, "srt_calibrate");
    double sum = 0, mx = 0, mn = 1e300;
    for (unsigned long long v : h) { sum += (double)v; mx = std::max(mx, (double)v); mn = std::min(mn, (double)v); }
    memset(out, 0, sizeof(*out));
    out->wave_cycles_mean = sum / n_waves; out->wave_cycles_max = mx; out->wave_cycles_min = mn; out->wall_ms = ms;
    out->instr_per_wave = (uint64_t)iters * 32u;
    out->n_waves = n_waves; out->n_cu = n_blocks; out->waves_per_simd = waves_per_simd;
    return SRT_OK;
}

}  // extern "C"
