// srt_internal.h -- shared between the C-ABI implementation (srt_capi.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

struct srt_ctx;
extern "C" int srt_internal_init_device_params(srt_ctx *c, uint32_t tx, uint32_t ty, uint32_t bx, uint32_t by, uint32_t chunk_w, uint32_t chunk_h,
                                               uint32_t spp, uint32_t bounce_limit, uint64_t seed, int wait);      // not exported (hidden visibility)

extern "C" uint32_t srt_internal_gather_planes(const srt_ctx *c);      // planes of the context's exchange unit (3 or 9); not exported

namespace srt {

constexpr int kTilePlanes = 9;      // quantised rgb | unquantised sRGB | XYZ sums
constexpr int kTileGroups = 3;      // ... in three groups of three planes; group 0 (the quantised framebuffer) is what the multi-GPU gather moves
constexpr int kGroupPlanes = 3;
constexpr int kTileLanes = 64;      // one wave = one 8x8 pixel tile
constexpr int kCounters = 32;       // rays, node_visits, tri_tests, box_tests, utilisation counters (instrumented build); [23] queue invariant, [24] hits

// Kernel arguments of one render launch.  All pointers are device pointers.
struct RenderParams {
    // scene (HBM layout: DESIGN.md)
    const float4 *nodes;       // 4 float4 per INNER record (records [0, n_inner))
    const float *nodes_sw;     // 20 floats per INNER record, pre-swizzled (srt_host.cpp, flatten_scene); null when the tree is LDS resident
    const float4 *fringe;      // 6 float4 per FRINGE record (records [n_inner, n_records)), triangle data inline
    const float4 *tris;        // 3 float4 per triangle
    const float2 *mat_sd;      // per material 96 pairs (94 used): (sd[k], sd[k+1]); table n_materials = the background
    const float4 *shade;       // 3 float4 per triangle: {n, bits(mat)} {bits(type), fuzz, B0, B1} {B2, C0, C1, C2}
    const float4 *mat_par;     // per material 2 float4: {bits(type), fuzz, B0, B1}, {B2, C0, C1, C2}
    const float4 *cmf;         // 96 rows (95 used): { x_bar, y_bar, z_bar, D65n }
    int root_ref;              // >= 0 record index, < 0: ~triangle (single-leaf tree)
    int stack_depth;           // LDS stack entries per lane
    int n_inner;               // records below this index have two internal children (box tests only)
    int n_cached;              // records below this index are resident in LDS (top of the tree)
    int n_records;             // number of paired-child records
    uint32_t fringe_stride;    // bytes between two FRINGE records: 96 (packed) or 128 (one cache line each: trees served by L2)
    uint32_t n_materials;
    uint32_t n_tris;
    uint32_t paired;           // host side only (the launcher's choice of variant): the tree has no node with exactly one leaf child
    // camera_data (rendering/rendering.cuh:28-36)
    float du[3], dv[3], p00[3];
    float defocus_angle;
    float center[3], disk_u[3], disk_v[3];
    // launch geometry
    uint32_t width, height, offx, offy;   // chunk (rendering.cu:153)
    uint32_t tx, ty, bx, by;              // the reference's block / grid dims: define idx and the RNG seed
    uint32_t spp, bounce_limit;
    uint32_t tiles_x, tiles_y, n_tiles;   // 8x8 tiles covering the chunk
    uint32_t rank, world;                 // this launch renders tiles t with t % world == rank
    uint32_t tiles_local;                 // number of tiles this rank owns (pixel queue length / 64)
    uint32_t *pixel_counter;              // device word, zeroed before the launch: head of the pixel queue
    const uint32_t *tile_order;           // optional: queue slot -> local tile (cost-descending order); null = identity
    uint32_t *tile_cost;                  // probe mode: [0, tiles_local) per local tile, node records visited by its pixels; [tiles_local, 2 tiles_local) its most expensive pixel
    const uint32_t *queue_rows;           // optional: [0] = number of queue rows, [1] = the largest tile cost (device-written by order_tiles_kernel)
    const uint32_t *prio_cost;            // optional: the probe's per-tile cost, read by the render launch for its wave priorities
    uint32_t queue_rows_bound;            // host-side upper bound of the row count (= tiles_local without splitting)
    uint32_t waves_per_cu_override;       // 0 = occupancy API
    uint32_t score_shade, score_fringe;   // step-choice weights, 256 / relative step cost (both >= 1)
    uint32_t debug_lane_limit;            // experiments only (env SRT_DEBUG_LANE_LIMIT): lanes >= limit of every tile stay idle
    // state / outputs
    uint32_t *rng;                        // SoA: 6 planes of n_lanes words, indexed by the block-linear idx
    uint32_t n_lanes;                     // tx*ty*bx*by
    float *tile_out;                      // [group][local tile (tiles_padded of them)][plane of the group][lane]
    uint32_t tile_group_stride;           // floats between two groups = tiles_padded * 3 * 64
    uint32_t write_parity;                // 0: only group 0 (the reference's framebuffer values) is written; 1: + the two parity groups
    unsigned long long *counters;
    uint32_t *wave_debug;                 // instrumented build, optional: OrderProfile header, then 4 words per wave (see srt_get_wave_debug)
};

// Child-order profile of an instrumented launch (srt_order_children_by_profile): the pointers travel in a header IN FRONT of the
// per-wave words of RenderParams::wave_debug (this struct, then 4 words per launched wave) so that RenderParams -- the kernel argument
// of every variant -- stays as it is and the header's place does not depend on the number of waves a launch starts.  magic == kOrderProfileMagic marks a launch that collects the profile.
struct OrderProfile {
    unsigned long long magic;
    const int32_t *leaf;       // per triangle: node index of its leaf
    const int32_t *up;         // per node: parent node * 2 + (1 if the node is the right child), -1 for the root
    const float *sibbox;       // per node: the box of its sibling (xmin xmax ymin ymax zmin zmax)
    uint32_t *cnt;             // [parent * 2 + side]: closest hits found under that child while the sibling's box lay on the ray beyond the hit
    unsigned long long n_nodes;
};
constexpr unsigned long long kOrderProfileMagic = 0x5352544f52444552ull;

struct ScatterParams {
    const float *gathered;     // [rank][group (groups of them)][tiles_padded][plane of the group][lane]
    uint32_t groups;           // 1: only the quantised framebuffer was gathered (12 B / pixel); 3: the parity planes too
    float *fb[9];              // block-linear planes: r g b | lin r g b | X Y Z
    uint32_t width, height;
    uint32_t tx, ty, bx, by;
    uint32_t tiles_x, n_tiles, world, tiles_padded;
};

hipError_t launch_init_rng(uint32_t *rng, uint32_t n_lanes, uint64_t seed, hipStream_t st);
// Test knobs of a context (srt_set_test_knobs; from the environment only under SRT_TEST_KNOBS=1, read once at srt_create): they pick
// the kernel variant / cache size a launch plan would not pick by itself, so that every instantiated variant can be held to the CPU oracle by the tests.
struct PlanKnobs { bool wide_refs = false; int lds_cache_max = -1; };
// mode 0 render, 1 instrumented, 2 cost probe; waves_launched (optional) = persistent waves of the launch
hipError_t launch_render(const RenderParams &p, const PlanKnobs &knobs, uint32_t n_cu, int mode, hipStream_t st, uint32_t *waves_launched = nullptr);
hipError_t launch_order_tiles(const uint32_t *cost, uint32_t *sorted, uint32_t *rows, uint32_t n, uint32_t n_waves,
                              uint32_t split_load_pct, uint32_t *queue_info, uint32_t order_max_pct, hipStream_t st);
hipError_t launch_scatter(const ScatterParams &p, hipStream_t st);
hipError_t launch_unswizzle(const float *const src[3], float *const dst[3], uint32_t tx, uint32_t ty, uint32_t bx, uint32_t by,
                            uint32_t n_cols, uint32_t n_rows, uint32_t offx, uint32_t offy, uint32_t image_width,
                            uint32_t image_height, hipStream_t st);
hipError_t launch_trace(const RenderParams &p, const float *rays, size_t n, float *out, hipStream_t st);
hipError_t launch_op_sweep(int which, const float *a, const float *b, size_t n, float *out, hipStream_t st);
hipError_t launch_calib(int kind, uint32_t n_blocks, uint32_t threads, uint32_t iters, float *sink, unsigned long long *cycles, const float4 *table,
                        uint32_t n_records, hipStream_t st);
int calib_kinds();
bool render_narrow_refs(int n_records, const PlanKnobs &k);
bool render_paired_variant(bool tree_is_paired, bool narrow, bool all_cached);      // does launch_render pick render_kernel<.., PAIRED = true>?
size_t render_lds_bytes(int stack_depth, int waves_per_block, int n_cached, int n_records, const PlanKnobs &k);
void render_launch_shape(int stack_depth, int n_records, int n_inner, const PlanKnobs &k, int &waves_per_block, int &n_cached);
struct LaunchPlan { int waves_per_block, blocks_per_cu, waves_per_cu, waves_per_eu, n_cached; bool all_cached; };
void render_launch_plan(int stack_depth, int n_records, int n_inner, const PlanKnobs &k, LaunchPlan &lp);   // what launch_render will do for this scene

}  // namespace srt
