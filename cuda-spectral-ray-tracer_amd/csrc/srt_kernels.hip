// srt_kernels.hip -- hand-written gfx950 kernels of the spectral path tracer.
//
// render_kernel is the whole hot path of the reference's spectral_render_kernel
// (rendering/rendering.cu:151-235).  A lane owns one pixel at a time and consumes that pixel's XORWOW
// stream in the reference's order (samples and bounces looped inside the lane); a wave pulls pixels
// from a cost-ordered queue (8x8 tiles, split into narrower rows when a launch is chain-bound).
// Unlike the reference (one thread per pixel walking a pointer tree with a 64-entry local-memory stack
// and a 36-byte hit record in shared memory) the lane is a small state machine and the wave a scheduler:
// shading passes (scatter, path end, new camera ray) alternate with branch-free traversal steps of two
// kinds -- INNER records (two child boxes, the whole inner tree LDS resident when it fits) and FRINGE
// records (a leaf child: box + triangle tests from one 96-byte record of (left, right) pairs) -- and
// the wave serves whichever kind has the most waiting lanes per unit of cost.  The traversal stack lives
// in LDS (one column per lane, lane-interleaved, newest entry in a register) and the hit record is just
// (t, triangle).  DESIGN.md section 5 has the measurements behind every one of these choices.
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
#include <algorithm>
#include <stdlib.h>

#include "srt_color_consts.h"
#include "srt_kernel_common.h"

namespace srt {

#ifndef SRT_INNER_BURST
#define SRT_INNER_BURST 8
#endif
#ifndef SRT_BURST_DROP
#define SRT_BURST_DROP 3
#endif
#ifndef SRT_ASM_BURST
#define SRT_ASM_BURST 2      /* 0: C++ bursts, 1: assembly bursts, 2: assembly decision + bursts */
#endif
#ifndef SRT_INNER_BURST_L2
#define SRT_INNER_BURST_L2 4
#endif
#ifndef SRT_ASM_BURST_L2
#define SRT_ASM_BURST_L2 1   /* 32-bit references, inner tree partly in L2 (cfg 5's mesh): INNER bursts in assembly (inner_burst4_mixed_asm) */
#endif
#ifndef SRT_BURST_DROP_L2
#define SRT_BURST_DROP_L2 2
#endif
// ... when part of the inner tree comes from L2 (ALL_CACHED false) a visit is twice as long and shorter bursts that end at half of
// their lanes win: 4 / 2 (sweep on cfg 5's scene, 64 spp: 8/3 238.3, 4/3 230.8, 12/3 244.6, 8/2 231.8, 8/5 243.3, 2/2 237.2, 3/2 234.8,
// 4/2 229.9, 6/2 230.4, 4/1 246.0 ms)
constexpr int kInnerBurstL2 = SRT_INNER_BURST_L2;
constexpr uint32_t kBurstDropL2 = SRT_BURST_DROP_L2;
constexpr int kInnerBurst = SRT_INNER_BURST;   // at most this many inner steps between two scheduling decisions (fully unrolled)
constexpr uint32_t kBurstDrop = SRT_BURST_DROP;   // ... and the burst ends once fewer than 1 / kBurstDrop of its lanes are still at inner records
static inline size_t round16(size_t v) { return (v + 15) & ~(size_t)15; }
// 16-bit child references and stack entries when the record indices fit 15 bits (a 16-bit stack slot also holds the sentinel -1).
// PlanKnobs (test knobs of a context, srt_set_test_knobs -- never the environment of a launch): wide_refs sends small trees through the
// 32-bit variants as well, lds_cache_max caps the LDS-resident inner records.
bool render_narrow_refs(int n_records, const PlanKnobs &k) { return !k.wide_refs && n_records <= 32767; }
static inline size_t cache_bytes(int n_cached, bool narrow) { return round16((size_t)n_cached * (narrow ? 52 : 56)); }
static inline size_t stack_slots(int stack_depth) { return (size_t)(stack_depth < 1 ? 1 : stack_depth) + kStackSentinels; }
static inline size_t stack_bytes(int stack_depth, bool narrow) { return stack_slots(stack_depth) * 64 * (narrow ? 2 : 4); }
size_t render_lds_bytes(int stack_depth, int waves_per_block, int n_cached, int n_records, const PlanKnobs &k) {
    const bool narrow = render_narrow_refs(n_records, k);
    return (size_t)kLdsTablesF4 * 16 + cache_bytes(n_cached, narrow) + (size_t)waves_per_block * stack_bytes(stack_depth, narrow);
}
// One workgroup per CU when the stacks leave room for a useful cache: 16 waves share it.  Deep trees (big stacks)
// fall back to smaller groups.  Only INNER records are cached (n_inner of them, breadth-first order).
void render_launch_shape(int stack_depth, int n_records, int n_inner, const PlanKnobs &k, int &waves_per_block, int &n_cached) {
    const bool narrow = render_narrow_refs(n_records, k);
    const size_t stack = stack_bytes(stack_depth, narrow);
    waves_per_block = 16;
    while (waves_per_block > 1 && (size_t)kLdsTablesF4 * 16 + waves_per_block * stack + 16 * 1024 > kLdsBudget) waves_per_block /= 2;
    const size_t blocks_per_cu = 16 / waves_per_block;
    const size_t per_block = kLdsBudget / blocks_per_cu;
    const size_t fixed = (size_t)kLdsTablesF4 * 16 + waves_per_block * stack + 64;
    size_t room = per_block > fixed ? (per_block - fixed) / (narrow ? 52 : 56) : 0;
    if (room > (size_t)n_inner) room = (size_t)n_inner;
    if (k.lds_cache_max >= 0) room = std::min(room, (size_t)k.lds_cache_max);
    n_cached = (int)room;
}
// The launch plan of a scene: one 16-wave workgroup per CU (4 waves / SIMD at this kernel's 126 VGPRs) whenever the stacks leave
// room for a useful cache; ALL_CACHED when the whole inner tree fits it.  (Round 3 measured a second shape for trees that do not
// fit -- 256-thread workgroups, five per CU, 96 VGPRs, five waves per SIMD -- at -5 % on the 100k-triangle mesh and dropped it:
// profiles/r03/experiments/cfg5_negative_results.txt.)
void render_launch_plan(int stack_depth, int n_records, int n_inner, const PlanKnobs &k, LaunchPlan &lp) {
    render_launch_shape(stack_depth, n_records, n_inner, k, lp.waves_per_block, lp.n_cached);
    lp.all_cached = lp.n_cached == n_inner;
    lp.waves_per_eu = 4;
    lp.blocks_per_cu = 16 / lp.waves_per_block;
    lp.waves_per_cu = lp.waves_per_block * lp.blocks_per_cu;
}

#ifndef SRT_PAIRED_VARIANT
#define SRT_PAIRED_VARIANT 1      /* 0: kernel experiments -- paired trees run the general variant too */
#endif
// (instantiated for the two shapes real scenes launch: 16-bit references with the whole inner tree in LDS, 32-bit references with the
// inner tree partly in L2; the mid-size shape <.,1,0> and the test-only shape <.,0,1> run the general variant)
bool render_paired_variant(bool tree_is_paired, bool narrow, bool all_cached) { return SRT_PAIRED_VARIANT != 0 && tree_is_paired && narrow == all_cached; }

// Cost band (0 = most expensive) of the first queue row that wave w of a workgroup takes (render_kernel S3): row = band * number of
// workgroups + workgroup.  SRT_ASSIGN_PERM picks how the bands are dealt to the waves of a workgroup: 0 in wave order, 1 transposed
// (w % 4) * 4 + w / 4, 2 boustrophedon over groups of four (0 1 2 3 | 7 6 5 4 | 8 ...), so that the four waves w, w + 4, w + 8, w + 12
// -- one SIMD's, if waves go to SIMDs round robin -- hold bands of equal total rank.
#ifndef SRT_ASSIGN_PERM
#define SRT_ASSIGN_PERM 0
#endif
__device__ __forceinline__ uint32_t assigned_band(uint32_t w) {
    if (SRT_ASSIGN_PERM == 1) return (w & 3u) * 4u + (w >> 2);
    if (SRT_ASSIGN_PERM == 2) return (w & 4u) ? ((w & ~3u) | (3u - (w & 3u))) : w;
    return w;
}

// init_random_states (rendering.cu:120-138): curand_init(seed + idx, 0, 0)
__global__ void init_rng_kernel(uint32_t *rng, uint32_t n_lanes, uint64_t seed) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_lanes) return;
    Rng s;
    rng_seed(s, seed + idx);
    rng[0 * (size_t)n_lanes + idx] = s.d;
    rng[1 * (size_t)n_lanes + idx] = s.v0;
    rng[2 * (size_t)n_lanes + idx] = s.v1;
    rng[3 * (size_t)n_lanes + idx] = s.v2;
    rng[4 * (size_t)n_lanes + idx] = s.v3;
    rng[5 * (size_t)n_lanes + idx] = s.v4;
}


// One lane = a small state machine that owns one pixel at a time:
//   TRAV   (tv.node >= 0)                 walking the BVH for its current ray
//   RESULT (tv.node == kTravDone)         traversal finished, hit record waits to be shaded
//   IDLE   (!have_path)                   needs a new camera ray, or a new pixel when its samples are used up
//   DEAD                                  pixel queue drained
//   PARKED                                split queue rows only: waits until the wave's expensive pixels are done
// The wave alternates between traversal steps (INNER or FRINGE: a step serves the lanes that sit at that kind of record)
// and shading passes (RESULT / IDLE lanes are shaded, regenerated and re-armed); what it does next is the kind of work
// with the most waiting lanes per unit of cost, so slow rays do not hold the other lanes hostage.  Pixels are handed
// out by one wave-aggregated atomic per refill; which lane renders a pixel has no influence on the result (the RNG
// stream belongs to the pixel).
// MODE 0: production; MODE 1: instrumented (counts V / T / utilisation); MODE 2: cost probe -- renders P.spp samples per
// pixel from a COPY of the RNG state, writes nothing but the per-tile traversal cost used to order the pixel queue.
// ALL_CACHED: the whole inner tree fits the LDS cache (n_cached == n_inner): the INNER step has no global fall-back path.
// PAIRED (instantiated for <., 1, 1> and <., 0, 0>): every FRINGE record holds two triangles (srt_scene_is_paired): the visit has no box test.
template <int MODE, bool NARROW, bool ALL_CACHED, bool PAIRED>
__global__ __launch_bounds__(1024) void render_kernel(const RenderParams P) {
    constexpr bool COUNT = (MODE == 1);
    constexpr bool PROBE = (MODE == 2);
    constexpr bool ITERS = COUNT || PROBE;
    extern __shared__ float4 lds4[];
    lds_uniforms *U = (lds_uniforms *)lds4;
    float4 *s_cmf = lds4 + kLdsUniF4;
    constexpr bool narrow = NARROW;           // P.n_records <= 65535 (launcher): 16-bit child references and stack entries
    const uint32_t nc = (uint32_t)P.n_cached;
    float4 *s_q0 = lds4 + kLdsTablesF4, *s_q1 = s_q0 + nc, *s_q2 = s_q1 + nc;
    uint32_t *s_r0 = reinterpret_cast<uint32_t *>(s_q2 + nc), *s_r1 = s_r0 + nc;
    const size_t cache_b = (((size_t)nc * (narrow ? 52u : 56u)) + 15u) & ~(size_t)15u;
    char *s_stack_base = reinterpret_cast<char *>(lds4 + kLdsTablesF4) + cache_b;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;

    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; a++) {
            U->du[a] = P.du[a]; U->dv[a] = P.dv[a]; U->p00[a] = P.p00[a]; U->center[a] = P.center[a];
            U->disk_u[a] = P.disk_u[a]; U->disk_v[a] = P.disk_v[a];
        }
        U->defocus_angle = P.defocus_angle;
        U->n_rows = P.queue_rows ? P.queue_rows[0] : P.tiles_local;   // written by order_tiles_kernel earlier on this stream
        U->lane_limit = P.debug_lane_limit ? P.debug_lane_limit : 64u;
        U->width = P.width; U->height = P.height; U->offx = P.offx; U->offy = P.offy;
        U->tx = P.tx; U->ty = P.ty; U->bx = P.bx; U->by = P.by;
        U->tiles_x = P.tiles_x; U->n_tiles = P.n_tiles; U->rank = P.rank; U->world = P.world; U->spp = P.spp; U->n_lanes = P.n_lanes;
        split_ptr(P.rng, U->rng); split_ptr(P.tile_out, U->tile_out); split_ptr(P.tile_order, U->tile_order);
        split_ptr(P.tile_cost, U->tile_cost); split_ptr(P.pixel_counter, U->pixel_counter);
        U->tile_group_stride = P.tile_group_stride;
        U->n_assigned_slots = gridDim.x * (blockDim.x >> 6) * 64u;      // one assigned first row per launched wave
        U->first_row_taken = 0u;
        split_ptr(P.prio_cost, U->prio_cost);
        U->prio_full = __float_as_uint(P.queue_rows && P.prio_cost ? (float)P.queue_rows[1] * (float)P.spp : 0.f);
    }
    for (uint32_t k = threadIdx.x; k < kLdsCmfF4; k += blockDim.x) s_cmf[k] = P.cmf[k];
    for (uint32_t k = threadIdx.x; k < nc; k += blockDim.x) {
        s_q0[k] = P.nodes[4 * k + 0]; s_q1[k] = P.nodes[4 * k + 1]; s_q2[k] = P.nodes[4 * k + 2];
        const float4 q3 = P.nodes[4 * k + 3];
        const uint32_t lr = __float_as_uint(q3.x), rr = __float_as_uint(q3.y);
        if (narrow) s_r0[k] = (lr & 0xffffu) | (rr << 16);
        else { s_r0[k] = lr; s_r1[k] = rr; }
    }
    __syncthreads();      // the only barrier: from here on every wave runs its own state machine

    NodeSrc ns;
    ns.global_nodes = make_rsrc(P.nodes, (uint32_t)P.n_inner * 64u);
    // (a tree that fits LDS always has packed 96-byte FRINGE records -- srt_upload_scene -- so that variant keeps the literal)
    ns.fringe_stride = ALL_CACHED ? 96u : P.fringe_stride;
    ns.global_fringe = make_rsrc(P.fringe, (uint32_t)(P.n_records - P.n_inner) * ns.fringe_stride);
    ns.global_nodes_sw = make_rsrc(P.nodes_sw, P.nodes_sw ? (uint32_t)P.n_inner * 80u : 0u);
    ns.n_inner = P.n_inner;
    ns.lds_q0 = (lds_cf4 *)s_q0; ns.lds_q1 = (lds_cf4 *)s_q1; ns.lds_q2 = (lds_cf4 *)s_q2;
    ns.lds_r0 = (lds_cu32 *)s_r0; ns.lds_r1 = (lds_cu32 *)s_r1;
    ns.n_cached = P.n_cached;
    const buf_rsrc shade_rsrc = make_rsrc(P.shade, P.n_tris * 48u);
    const buf_rsrc sd_rsrc = make_rsrc(P.mat_sd, (P.n_materials + 1u) * 768u);
    const uint32_t spp = P.spp;
    // the step-choice weights as opaque register values: read from the kernel arguments once, not re-loaded (an s_load and a
    // wait) every time the traversal loop comes round
    uint32_t w_shade = P.score_shade, w_fringe = P.score_fringe;
    asm volatile("" : "+s"(w_shade), "+s"(w_fringe));
    StackRef my_stack;
    {
        const size_t slots = (size_t)(P.stack_depth < 1 ? 1 : P.stack_depth) + kStackSentinels;
        my_stack.s16 = (lds_i16 *)(s_stack_base + (size_t)wave * slots * 128u) + lane;
        my_stack.s32 = (lds_i32 *)(s_stack_base + (size_t)wave * slots * 256u) + lane;
        stack_init<NARROW>(my_stack);      // this wave's own column: no barrier needed
    }

    // ---- lane state ---------------------------------------------------------------------------------------
    bool dead = false, have_path = false, have_pixel = false;
    uint32_t idx = 0;                       // block-linear index of the current pixel (RNG / framebuffer slot)
    uint32_t out_slot = 0;                  // tile_local * 192 + lane_in_tile (slot inside a plane group of the tile buffer)
    uint32_t pixel_ij = 0;                  // chunk-relative column | row << 16 (both are 16-bit quantities, Q17)
    Rng rs; rs.d = rs.v0 = rs.v1 = rs.v2 = rs.v3 = rs.v4 = 0u;
    V3 acc = mk(0.f, 0.f, 0.f);             // pixel_color (rendering.cu:212)
    uint32_t sample = 0, bounce = 0, valid = 0;
    V3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 1.f), inv = mk(0.f, 0.f, 1.f);
    float hero = kLambdaMin;
    float pw[kWavelengths];
#pragma unroll
    for (int k = 0; k < kWavelengths; k++) pw[k] = 0.f;
    Trav tv; tv.node = kTravIdle; tv.sp = 0u; tv.top = -1; tv.c = kFltMax; tv.hit = -1; tv.nf[0] = tv.nf[1] = tv.nf[2] = 0u;
    uint32_t n_rays = 0;
    TravStats ts;
    // Rows of an expensive tile that was split over several waves (see order_tiles_kernel) are exclusive: the lanes whose
    // slot is not part of the row's share, and the lanes that finish early, stay PARKED until every pixel of the wave is
    // done -- a wave that refilled itself with other pixels would slow the expensive chains down again.
    bool parked = false, exclusive = false;
    OrderProfile prof;      // instrumented build only: see srt_internal.h
    prof.magic = 0;
    if (COUNT && P.wave_debug) prof = *reinterpret_cast<const OrderProfile *>(P.wave_debug);      // (fixed place: in front of the per-wave words)
    unsigned long long t_shade = 0, t_inner = 0, t_fringe = 0, t_mark = 0;   // instrumented build: wave cycles per phase
    if (COUNT) t_mark = __builtin_amdgcn_s_memtime();
    const unsigned long long t_born = t_mark;
    unsigned long long t_dry = 0;            // instrumented build: when the pixel queue first came back empty for this wave
    uint32_t cur_tile_local = 0, pixel_iters0 = 0;   // probe / instrumented builds
    uint32_t pixel_rays0 = 0, max_pix_iters = 0, max_pix_rays = 0;
    uint32_t w_shade_passes = 0, l_shade = 0, l_cam = 0, w_reject_iters = 0;   // instrumented build: shading-phase occupancy
    uint32_t n_hits = 0;                     // instrumented build: queries that found a triangle (one shading record each)

    for (;;) {
        // parked lanes wake up when no lane of the wave holds a pixel any more
        if (__ballot(have_pixel) == 0ull) parked = false;
        // =========================== shading phase ===========================================================
        // every lane that is neither traversing nor dead nor parked goes through: shade -> path end -> pixel switch ->
        // camera ray -> start traversal.  ONE pass per turn of the outer loop: in the corner cases where a pass leaves lanes
        // without a query (bounce_limit 0, leaf-root BVH, NaN directions) the traversal phase below finds nothing to do and
        // the outer loop comes back here.
        if (__ballot(!dead && !parked && tv.node < 0) != 0ull) {
            bool end_path = false, begin_trav = false;
            if (COUNT) { w_shade_passes++; l_shade += (uint32_t)__popcll(__ballot(!dead && tv.node == kTravDone)); }
            float xyz_x = 0.0f, xyz_y = 0.0f, xyz_z = 0.0f;      // XYZ of a path that ends in this pass

            // ---- S1: shade a finished closest-hit query: one iteration of ray_bounce's loop (rendering.cu:22-36)
            if (!dead && tv.node == kTravDone) {
                tv.node = kTravIdle;
                float wl[kWavelengths];
                hero_expand(hero, wl);
                // the spectrum this segment multiplies into the path: the background on a miss (rendering.cu:24-27), the hit
                // material's reflectance otherwise (material.cu:95) -- one look-up site for both: table n_materials is the background
                uint32_t sd_table = P.n_materials;
                bool was_hit = false, hit_scattered = false;
                if (tv.hit < 0) {
                    end_path = true;        // miss: r.mul_spectrum(background) and stop
                } else {
                    if (COUNT && P.wave_debug && prof.magic == kOrderProfileMagic) {
                        // child-order profile: walk from the hit triangle's leaf to the root; at every ancestor whose OTHER child's box
                        // lies on the ray beyond the hit, note under which child the hit lay (visited first, that side prunes the other)
                        int k = prof.leaf[tv.hit];
                        for (int guard = 0; guard < 96; guard++) {
                            const int up = prof.up[k];
                            if (up < 0) break;
                            const float *b = prof.sibbox + 6 * (size_t)k;
                            const float x0 = (b[0] - ro.x) * inv.x, x1 = (b[1] - ro.x) * inv.x;
                            const float y0 = (b[2] - ro.y) * inv.y, y1 = (b[3] - ro.y) * inv.y;
                            const float z0 = (b[4] - ro.z) * inv.z, z1 = (b[5] - ro.z) * inv.z;
                            const float e = fmaxf(fmaxf(fmaxf(0.0f, fminf(x0, x1)), fminf(y0, y1)), fminf(z0, z1));
                            const float m = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                            if (e <= m && e > tv.c) atomicAdd(&prof.cnt[up], 1u);
                            k = up >> 1;
                        }
                    }
                    // rebuild the hit record from (t, triangle): tri::hit tail (tri.cu:36-39) + set_face_normal.  ONE round of
                    // loads: the triangle's shading record carries the normal, the material index and the material scalars.
                    const uint32_t srec = __umul24((uint32_t)tv.hit, 48u);      // (full-rate 24-bit multiply; < 2^24 triangles)
                    const f4v s0 = buf_load16(shade_rsrc, srec), s1 = buf_load16(shade_rsrc, srec + 16u), s2 = buf_load16(shade_rsrc, srec + 32u);
                    const V3 n_geo = mk(s0.x, s0.y, s0.z);
                    const V3 hp = ro + tv.c * rd;                                          // ray::at, ray.cuh:31-34
                    const bool front_face = dot(rd, n_geo) < 0;                            // hit_record.cuh:41
                    const V3 n = front_face ? n_geo : -n_geo;
                    const uint32_t mat = __float_as_uint(s0.w);
                    const uint32_t mtype = __float_as_uint(s1.x);
                    const float fuzz = s1.y;

                    // material::scatter (materials/material.cu:55-100)
                    V3 scatter_direction = mk(0.f, 0.f, 0.f);
                    float eps_sign = 1.0f;
                    bool did_scatter = true;
                    const V3 unit_in = unit_vector(rd);
                    if (mtype == 4u) {                                                      // EMISSIVE, :83-86
                        did_scatter = false;
                    } else if (mtype == 2u) {                                               // DIELECTRIC, :73-80
                        float ir = sellmeier_index(s1.z, s1.w, s2.x, s2.y, s2.z, s2.w, wl[0]);
                        // refraction_scatter, :102-136
                        float refraction_ratio = front_face ? (1.0f / ir) : ir;
                        float cos_theta = fminf(dot(-unit_in, n), 1.0f);
                        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
                        bool cannot_refract = refraction_ratio * sin_theta > 1.0f;
                        if (!cannot_refract)                                                // short-circuit ||, :114
                            cannot_refract = reflectance(cos_theta, refraction_ratio) > rng_uniform(rs);
                        if (cannot_refract) {
                            scatter_direction = reflect(unit_in, n);
                        } else {
                            scatter_direction = refract(unit_in, n, refraction_ratio);
                            eps_sign = -1.0f;
                            valid = 1;                                                      // :78-79 (Q6)
                        }
                    } else {
                        // METALLIC (:64-71) and LAMBERTIAN/default (:88-92) both start with random_unit_vector
                        const V3 ruv = random_unit_vector(rs);                            // vec3.cuh:221-227
                        if (mtype == 1u) {
                            V3 reflected = reflect(unit_in, n);                            // reflection_scatter, :22-37
                            scatter_direction = reflected + fuzz * ruv;
                            did_scatter = dot(scatter_direction, n) > 0;
                            if (!did_scatter) valid = 0;
                        } else {
                            scatter_direction = n + ruv;                                   // lambertian_scatter, :8-19
                            if (near_zero(scatter_direction)) scatter_direction = n;
                        }
                    }
                    sd_table = mat;
                    ro = hp + (eps_sign * kEpsilon) * n;                                    // :96 (Q9)
                    rd = scatter_direction;                                                 // :97
                    hit_scattered = did_scatter;
                    was_hit = true;
                    if (COUNT) n_hits++;
                }
                // r_in.mul_spectrum(spectral_distribution) (:95), after valid_wavelengths was updated (Q8).  A path that ends
                // here (miss, or no scattered ray) is converted in the same pass: dev_spectrum_to_XYZ (color.cu:88-104) needs
                // the same interpolation coordinates.  (A path that ends because the bounce limit is reached contributes
                // nothing: valid_wavelengths = 0, rendering.cu:38.)
                //
                // All seven look-ups are issued together and the products are unconditional: power[k] for k >= valid is dead
                // (valid only shrinks inside a path and a new path resets all seven), so multiplying it does no harm, while a
                // per-wavelength `if (k < valid)` would put seven dependent memory round trips one after the other.
                const bool ends_here = !was_hit || !hit_scattered;
                const float delta_lambda = (kLambdaMax - kLambdaMin) / (float)kWavelengths;
                const uint32_t sd_base = __umul24(sd_table, 768u);
                int off[kWavelengths];
                float wgt[kWavelengths];
                float2 sp[kWavelengths];
#pragma unroll
                for (int k = 0; k < kWavelengths; k++) {
                    interp_coords(wl[k], off[k], wgt[k]);
                    sp[k] = buf_load8(sd_rsrc, sd_base + (uint32_t)off[k] * 8u);
                }
#pragma unroll
                for (int k = 0; k < kWavelengths; k++) pw[k] *= interp_pair(sp[k], wgt[k]);
                if (ends_here) {
#pragma unroll
                    for (int k = 0; k < kWavelengths; k++) {
                        const float4 r0 = s_cmf[off[k]], r1 = s_cmf[off[k] + 1];
                        // the reference sums the first `valid` terms; the others contribute +0, which leaves a sum unchanged: their POWER is
                        // replaced by +0 (one select per wavelength instead of three per term) -- the interpolated colour-matching value is a
                        // convex combination of non-negative table entries, so (value * +0) * delta = +0 exactly
                        const float w = wgt[k], power = (uint32_t)k < valid ? pw[k] : 0.0f;
                        xyz_x += ((1.0f - w) * r0.x + w * r1.x) * power * delta_lambda;
                        xyz_y += ((1.0f - w) * r0.y + w * r1.y) * power * delta_lambda;
                        xyz_z += ((1.0f - w) * r0.z + w * r1.z) * power * delta_lambda;
                    }
                }
                if (was_hit) {
                    if (!hit_scattered) {
                        end_path = true;
                    } else {
                        bounce++;
                        if (bounce < P.bounce_limit) begin_trav = true;
                        else { valid = 0; end_path = true; }                               // loop exhausted, :38 (Q7)
                    }
                }
            }

            // ---- S2: path end: pixel_color += dev_spectrum_to_XYZ(...) (rendering.cu:227, color.cu:88-104) --------
            if (end_path) {
                acc = acc + mk(xyz_x, xyz_y, xyz_z);
                have_path = false;
            }

            // ---- S3: pixel switch: all samples of the current pixel done (or no pixel yet) ------------------------
            const bool switched_any = SRT_PRIO_MODE != 0 && (ALL_CACHED || SRT_PRIO_L2) && __ballot(!dead && !parked && tv.node < 0 && !have_path && !begin_trav && (!have_pixel || sample == spp)) != 0ull;
            if (!dead && !parked && tv.node < 0 && !have_path && !begin_trav && (!have_pixel || sample == spp)) {
                if (have_pixel && PROBE) {
                    // cost of this pixel = node records it visited (+1 so that empty pixels still sort after real ones)
                    uint32_t *tcost = join_ptr<uint32_t>(U->tile_cost[0], U->tile_cost[1]);
                    atomicAdd(tcost + cur_tile_local, ts.n_iters - pixel_iters0 + 1u);
                    // ... and the tile's most expensive pixel (second half of the array): a pixel is one sequential chain, and a tile
                    // of average cost may hold a few very long ones (the silhouette of a glass object); order_tiles_kernel can rank by it
                    atomicMax(tcost + P.tiles_local + cur_tile_local, ts.n_iters - pixel_iters0 + 1u);
                    have_pixel = false;
                }
                if (have_pixel && COUNT) { max_pix_iters = max(max_pix_iters, ts.n_iters - pixel_iters0); max_pix_rays = max(max_pix_rays, n_rays - pixel_rays0); }
                if (have_pixel && !PROBE) {
                    // store RNG state (rendering.cu:232) and save_to_fb (rendering.cu:140-149)
                    {
                        uint32_t *rng = join_ptr<uint32_t>(U->rng[0], U->rng[1]);
                        const size_t nl = U->n_lanes;
                        rng[0 * nl + idx] = rs.d; rng[1 * nl + idx] = rs.v0; rng[2 * nl + idx] = rs.v1;
                        rng[3 * nl + idx] = rs.v2; rng[4 * nl + idx] = rs.v3; rng[5 * nl + idx] = rs.v4;
                    }
                    // pixel_color / float(spp) -> (1/spp) * v ; XYZ_to_sRGB (color.cu:35-41, vec3.cuh:80-91)
                    const float inv_spp = 1.0f / (float)spp;
                    const V3 c = inv_spp * acc;
                    const float r_lin = (SRT_XYZ2RGB_00 * c.x) + (SRT_XYZ2RGB_01 * c.y) + (SRT_XYZ2RGB_02 * c.z);
                    const float g_lin = (SRT_XYZ2RGB_10 * c.x) + (SRT_XYZ2RGB_11 * c.y) + (SRT_XYZ2RGB_12 * c.z);
                    const float b_lin = (SRT_XYZ2RGB_20 * c.x) + (SRT_XYZ2RGB_21 * c.y) + (SRT_XYZ2RGB_22 * c.z);
                    const float r = correct_channel(r_lin), g = correct_channel(g_lin), b = correct_channel(b_lin);
                    // three groups of three planes: the quantised framebuffer (what a multi-GPU gather moves), then the two
                    // parity groups (unquantised sRGB, XYZ sums)
                    float *o = join_ptr<float>(U->tile_out[0], U->tile_out[1]) + out_slot;
                    const size_t gs = U->tile_group_stride;
                    o[0 * kTileLanes] = (float)(int)(r * 255.99f);      // expand_sRGB (color.cu:43-49, Q15)
                    o[1 * kTileLanes] = (float)(int)(g * 255.99f);
                    o[2 * kTileLanes] = (float)(int)(b * 255.99f);
                    if (P.write_parity) {      // (only when the caller asked for the parity planes: srt_set_gather_planes(ctx, 9))
                        o[gs + 0 * kTileLanes] = r; o[gs + 1 * kTileLanes] = g; o[gs + 2 * kTileLanes] = b;
                        o[2 * gs + 0 * kTileLanes] = acc.x; o[2 * gs + 1 * kTileLanes] = acc.y; o[2 * gs + 2 * kTileLanes] = acc.z;
                    }
                    have_pixel = false;
                }
                // fetch the next pixel of this rank's queue (wave-aggregated atomic); skip slots outside the chunk
                bool searching = true;
                if (exclusive) { parked = true; exclusive = false; searching = false; }   // wait for the rest of the split row
                while (searching) {
                    const unsigned long long m = __ballot(1);
                    const int leader = __ffsll((long long)m) - 1;
                    uint32_t base = 0;
                    // The FIRST row of every wave is assigned, not raced for: wave w of workgroup b takes row w * (number of
                    // workgroups) + b of the cost-descending queue, so every CU -- and, with consecutive waves of a workgroup on
                    // different SIMDs, every SIMD -- starts with one row of each cost band instead of sixteen neighbours of the
                    // sorted order (the most expensive tiles of a launch used to share a handful of CUs for their whole life), and
                    // the placement no longer depends on the order in which the waves' first atomics arrive.  Later rows come
                    // from the shared counter, which starts behind the assigned ones.
                    // Invariant: a wave's FIRST fetch is made by all 64 lanes (every lane starts without a pixel, alive and not parked,
                    // and lane_limit is applied after the fetch), so `m == ~0` holds whenever the wave's bit is still clear and the
                    // assigned row is handed out whole.  The instrumented build checks it (counter 23, srt_get_stats fails on it): a
                    // partial first fetch would leave the assigned row unrendered.  (Each wave reads and sets only its own bit; the
                    // atomicOr of the other waves touch other bits of the word.)
                    const bool row_open = SRT_ASSIGN_FIRST_ROW != 0 && ((U->first_row_taken >> wave) & 1u) == 0u;
                    if (COUNT && row_open && m != ~0ull && lane == (uint32_t)leader) atomicAdd(&P.counters[23], 1ull);
                    const bool assigned = row_open && m == ~0ull;
                    if (assigned) {      // (the per-wave "taken" bit lives in LDS: nothing stays live in the persistent loop for it)
                        if (lane == 0) atomicOr((unsigned int *)&U->first_row_taken, 1u << wave);
                        base = (assigned_band(wave) * gridDim.x + blockIdx.x) * 64u;
                    }
                    else {
                        if ((int)lane == leader) base = atomicAdd(join_ptr<uint32_t>(U->pixel_counter[0], U->pixel_counter[1]), (uint32_t)__popcll(m));
                        base = (uint32_t)__shfl((int)base, leader, 64) + (SRT_ASSIGN_FIRST_ROW != 0 ? U->n_assigned_slots : 0u);
                    }
                    const uint32_t pix = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (pix >= U->n_rows * 64u) { dead = true; searching = false; if (COUNT && t_dry == 0) t_dry = __builtin_amdgcn_s_memtime(); }
                    else {
                        // queue row = tile | part << 22 | s << 28: the row covers lanes [part * (64 >> s), (part + 1) * (64 >> s))
                        // of the tile (s = 0: the whole tile).  Cost-descending order, see order_tiles_kernel.
                        const uint32_t *tile_order = join_ptr<const uint32_t>(U->tile_order[0], U->tile_order[1]);
                        // (no ordered queue: the row IS the local tile, whatever its magnitude -- nothing to decode)
                        const uint32_t row = tile_order ? tile_order[pix >> 6] : 0u;
                        const uint32_t tile_local = tile_order ? (row & 0x3fffffu) : (pix >> 6);
                        const uint32_t row_part = (row >> 22) & 63u, row_s = (row >> 28) & 7u;
                        const uint32_t lt = pix & 63u;
                        // A split row is meant for a wave that takes it whole (all 64 lanes fetch together: first fill, or
                        // after an exclusive row).  Then the lanes outside the row's share park.  A lane that refills on its own
                        // and lands in a split row just skips a slot that is not the row's share, and renders one that is like
                        // any other pixel.
                        const bool whole_wave = m == ~0ull;
                        if ((lt >> (6u - row_s)) != row_part) {
                            if (whole_wave) { parked = true; searching = false; }
                            continue;
                        }
                        exclusive = whole_wave && row_s != 0u;
                        const uint32_t tile = U->rank + U->world * tile_local;
                        const uint32_t tiles_x = U->tiles_x;
                        const uint32_t tile_x = tile % tiles_x, tile_y = tile / tiles_x;
                        const uint32_t i = tile_x * 8u + (lt & 7u);          // chunk-relative column (rendering.cu:156)
                        const uint32_t j = tile_y * 8u + (lt >> 3);          // chunk-relative row    (rendering.cu:157)
                        const uint32_t gtx = U->tx, gty = U->ty, gbx = U->bx;
                        // pixels outside the chunk (or the reference grid) never touch RNG or output (rendering.cu:205)
                        if ((tile < U->n_tiles) && (lt < U->lane_limit) && (i < U->width) && (j < U->height) && (i / gtx < gbx) && (j / gty < U->by)) {
                            idx = block_linear_idx(i, j, gtx, gty, gbx);
                            out_slot = tile_local * (uint32_t)(kGroupPlanes * kTileLanes) + lt;
                            pixel_ij = i | (j << 16);
                            {
                                const uint32_t *rng = join_ptr<const uint32_t>(U->rng[0], U->rng[1]);      // rendering.cu:209
                                const size_t nl = U->n_lanes;
                                rs.d = rng[0 * nl + idx]; rs.v0 = rng[1 * nl + idx]; rs.v1 = rng[2 * nl + idx];
                                rs.v2 = rng[3 * nl + idx]; rs.v3 = rng[4 * nl + idx]; rs.v4 = rng[5 * nl + idx];
                            }
                            acc = mk(0.f, 0.f, 0.f);
                            sample = 0;
                            if (PROBE) { cur_tile_local = tile_local; pixel_iters0 = ts.n_iters; }
                            if (COUNT) { pixel_iters0 = ts.n_iters; pixel_rays0 = n_rays; }
                            have_pixel = true;
                            searching = false;
                        }
                    }
                }
                // Drain this block's memory traffic here, once per pixel: its stores and generic-pointer (flat) accesses share
                // the vmcnt / lgkmcnt counters with the record loads of the traversal loop and complete out of order with
                // respect to them, so anything that MAY still be in flight makes the compiler wait for vmcnt(0) at the first
                // load result a FRINGE visit uses -- all six record loads before touching the first one.
                __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0)
            }

#if SRT_PRIO_MODE
            // ---- wave priority: least slack first ----------------------------------------------------------------------------
            // The SIMD's arbiter prefers OLDER waves (srt_calib: with four resident waves two run at their single-wave speed and
            // two starve) and persistent waves keep their age order for the whole launch: which pixels run fast is an accident of
            // the dispatch order, and the starved wave of every SIMD is left alone with its last pixels when the others have gone
            // (a lone wave reaches half of the SIMD's issue rate at best; tools/wave_tail.py).  Here a wave sets its own priority
            // from the longest chain it still holds: the probe's cost of a lane's tile x the samples the lane still has to draw,
            // against the same product for the most expensive tile of the launch -- more than 1/4, 1/16, 1/64 of it: priority 3, 2,
            // 1 (sweep: profiles/r03/experiments/wave_priorities.txt).  Lagging waves rise, waves that only hold cheap or nearly
            // finished pixels fall: expensive chains get the issue slots, and the waves of a SIMD finish together.  Re-evaluated
            // whenever a lane of the wave switched pixel.  Not compiled into the variant for trees served by L2: that one is
            // memory-bound, the priorities cost it 1.5-2 % on a full frame and return 2 % on one rank's share of an 8-rank frame.
            // Scheduling only: results do not depend on it.
            if (!PROBE && (ALL_CACHED || SRT_PRIO_L2) && switched_any) {
                float rem = 0.f;
                const uint32_t *tc = join_ptr<const uint32_t>(U->prio_cost[0], U->prio_cost[1]);
                if (tc && have_pixel && !dead) rem = (float)tc[out_slot / (uint32_t)(kGroupPlanes * kTileLanes)] * (float)(spp - sample);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) rem = fmaxf(rem, __shfl_xor(rem, off, 64));
                const float full = __uint_as_float(U->prio_full);
                const uint32_t pr = (uint32_t)__builtin_amdgcn_readfirstlane((int)((rem > (SRT_PRIO_T1 / 1024.0f) * full ? 1u : 0u) + (rem > (SRT_PRIO_T2 / 1024.0f) * full ? 1u : 0u) + (rem > (SRT_PRIO_T3 / 1024.0f) * full ? 1u : 0u)));
                if (pr == 0u) __builtin_amdgcn_s_setprio(0);
                else if (pr == 1u) __builtin_amdgcn_s_setprio(1);
                else if (pr == 2u) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(3);
            }
#endif

            // ---- S4: new camera ray: renderer::get_ray (rendering.cu:66-87) ----------------------------------------
            if (COUNT) l_cam += (uint32_t)__popcll(__ballot(!dead && tv.node < 0 && !have_path && !begin_trav && have_pixel && sample < spp));
            if (!dead && tv.node < 0 && !have_path && !begin_trav && have_pixel && sample < spp) {
                float px = -0.5f + rng_uniform(rs);                       // pixel_sample_square, :49-56
                float py = -0.5f + rng_uniform(rs);
                const V3 du = mk(U->du[0], U->du[1], U->du[2]), dv = mk(U->dv[0], U->dv[1], U->dv[2]);
                const V3 cam_center = mk(U->center[0], U->center[1], U->center[2]);
                // pixel_center = p00 + (float)i*du + (float)j*dv, i/j incl. the chunk offset (rendering.cu:76,221); recomputed per
                // camera ray from the packed pixel coordinates instead of living in three registers
                const V3 pixel_center = (mk(U->p00[0], U->p00[1], U->p00[2]) + (float)(U->offx + (pixel_ij & 0xffffu)) * du) + (float)(U->offy + (pixel_ij >> 16)) * dv;
                V3 pixel_sample = pixel_center + (px * du + py * dv);
                V3 origin = cam_center;
                if (!(U->defocus_angle <= 0.0f)) {                          // defocus_disk_sample, :42-47
                    float dx, dy;
#if SRT_ASM_RNG
                    rng_disk_loop_asm(rs, dx, dy);                         // random_in_unit_disk, vec3.cuh:240-246 (srt_device.h)
#else
                    for (;;) {                                             // random_in_unit_disk, vec3.cuh:240-246
                        dx = rng_pm1(rs);
                        dy = rng_pm1(rs);
                        if ((dx * dx + dy * dy) + 0.0f * 0.0f < 1.0f) break;
                    }
#endif
                    origin = (cam_center + dx * mk(U->disk_u[0], U->disk_u[1], U->disk_u[2])) +
                             dy * mk(U->disk_v[0], U->disk_v[1], U->disk_v[2]);
                }
                ro = origin;
                rd = pixel_sample - origin;                                // not normalised (Q10)
                hero = hero_draw(rs);                                      // ray ctor -> init_spectrum, ray.cuh:37-50
#pragma unroll
                for (int k = 0; k < kWavelengths; k++) pw[k] = 1.0f;
                valid = kWavelengths;
                sample++;
                bounce = 0;
                have_path = true;
                if (P.bounce_limit > 0) begin_trav = true;
                else { valid = 0; /* ray_bounce's loop body never runs (rendering.cu:22,38) */
                       have_path = false; /* contributes dev_spectrum_to_XYZ(valid = 0) = 0 */ }
            }

            // ---- S5: start the closest-hit query: bvh::hit(r, 0, FLT_MAX, rec, root) (rendering.cu:24) --------------
            if (begin_trav) {
                n_rays++;
                inv = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);           // aabb.cu:17, hoisted out of the box test
                ray_near_addresses(ns, inv, tv.nf);
                (void)trav_begin<ITERS>(tv, stack_base<NARROW>(my_stack), P.tris, P.root_ref, ro, rd, ts);   // a query that finishes at once leaves kTravDone
            }
        }

        if (COUNT) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_shade += now - t_mark; t_mark = now; }
        // =========================== traversal phase =========================================================
        // Two kinds of steps: INNER (record with two internal children: box tests only) and FRINGE (a leaf child:
        // box + triangle tests, several times the cost).  A step serves the lanes that sit at that kind of record.
        if (__ballot(!dead) == 0ull) break;
        const unsigned long long alive_mask = __ballot(!dead && !parked);
        if (alive_mask == 0ull) continue;     // only parked lanes left: wake them up
        const uint32_t n_alive = (uint32_t)__popcll(alive_mask);
        const uint32_t n_inner_u = (uint32_t)P.n_inner;
        if (!COUNT && n_alive == 1u) {
            // A wave with ONE working lane (a single-pixel row of a split tile, or the tail of the launch) has nothing to
            // schedule: walk the ray to the end in a tight loop.  Such waves are latency chains -- every instruction of the
            // step-choice logic is on the critical path of the pixel that bounds a chain-bound launch.
            while (__ballot(tv.node >= 0) != 0ull) {
                if (tv.node >= 0) {
                    if ((uint32_t)tv.node < n_inner_u) trav_step_inner<ITERS, NARROW, ALL_CACHED>(tv, ns, ro, inv, my_stack, ts);
                    else trav_step_fringe<ITERS, NARROW, PAIRED>(tv, ns, ro, rd, inv, my_stack, ts);
                }
            }
            continue;
        }
        if (!ITERS && NARROW && ALL_CACHED && kInnerBurst == 8 && kBurstDrop == 3u && SRT_ASM_BURST == 2) {
            // production build: the decision below and the INNER bursts it leads to are one assembly block (inner_phase_asm,
            // srt_device.h -- the same arithmetic); only the FRINGE visits come back here
            while (inner_phase_asm(tv, ns, ro, inv, n_inner_u, n_alive, w_shade, w_fringe) != 0u) {
                if (tv.node >= (int)n_inner_u) trav_step_fringe<ITERS, NARROW, PAIRED>(tv, ns, ro, rd, inv, my_stack, ts);
            }
        } else
        for (;;) {
            // Serve the kind of work with the most waiting lanes per unit of step cost (weights = 256 / relative cost of the
            // step, srt_capi.cpp).  Traversing lanes are alive (a lane retires or parks only between queries), so the lanes
            // waiting for a shading pass are n_alive - n_trav.
            const unsigned long long trav_mask = __ballot(tv.node >= 0);
            if (trav_mask == 0ull) break;
            const unsigned long long fringe_mask = __ballot(tv.node >= (int)n_inner_u);
            const uint32_t n_trav = (uint32_t)__popcll(trav_mask), n_fringe = (uint32_t)__popcll(fringe_mask);
            const uint32_t s_sh = (n_alive - n_trav) * P.score_shade;
            const uint32_t s_fr = n_fringe * P.score_fringe;
            const uint32_t s_in = (n_trav - n_fringe) << 8;
            if (s_sh > max(s_fr, s_in)) break;
            // (when every traversing lane sits at a fringe record, s_in = 0 < s_fr because the weight is >= 1: always progress)
            const bool do_fringe = s_fr > s_in;
            if (COUNT) { ts.w_iters++; ts.w_alive += n_alive; if (do_fringe) { ts.w_fringe++; ts.l_fringe += n_fringe; } else ts.l_inner += n_trav - n_fringe; }
            if (do_fringe) {
                if (tv.node >= (int)n_inner_u) trav_step_fringe<ITERS, NARROW, PAIRED>(tv, ns, ro, rd, inv, my_stack, ts);
                if (COUNT) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_fringe += now - t_mark; t_mark = now; }
            } else {
                // a short burst of inner steps between two scheduling decisions: the ballots / popcounts of the loop head
                // are a sizeable part of a 60-instruction step.  `at_inner` is carried across the back edge so that one
                // compare serves the step's EXEC mask and the loop exit.
                bool at_inner = (uint32_t)tv.node < n_inner_u;
                // the burst ends when fewer than `stay` lanes remain: ceil(lanes at the start / kBurstDrop), at least 1 -- so the
                // test also covers "no lane left" and costs a popcount and a compare per step
                constexpr int burst_len = ALL_CACHED ? kInnerBurst : kInnerBurstL2;
                constexpr uint32_t burst_drop = ALL_CACHED ? kBurstDrop : kBurstDropL2;
                const uint32_t stay = (n_trav - n_fringe + burst_drop - 1u) / burst_drop;
                if (!ITERS && NARROW && ALL_CACHED && kInnerBurst == 8 && SRT_ASM_BURST == 1) {
                    inner_burst8_asm(tv, ns, ro, inv, n_inner_u, stay);      // the same eight visits, hand-scheduled (srt_device.h)
                } else if (!ITERS && !NARROW && !ALL_CACHED && kInnerBurstL2 == 4 && SRT_ASM_BURST_L2 && P.nodes_sw != nullptr) {
                    inner_burst4_mixed_asm(tv, ns, ro, inv, n_inner_u, stay);      // LDS prefix + memory, hand-scheduled (srt_device.h)
                } else
#pragma unroll
                for (int burst = 0; burst < burst_len; burst++) {
                    if (at_inner) trav_step_inner<ITERS, NARROW, ALL_CACHED>(tv, ns, ro, inv, my_stack, ts);
                    at_inner = (uint32_t)tv.node < n_inner_u;
                    const unsigned long long m = __ballot(at_inner);
                    if (COUNT && burst > 0) { ts.w_iters++; ts.w_alive += n_alive; }
                    if (COUNT && burst < burst_len - 1) ts.l_inner += (uint32_t)__popcll(m);
                    // leave early once most of the lanes the burst started with have moved on (fringe record, finished query):
                    // the remaining few are better served together with the lanes a new decision brings in
                    // (bursts of 8: exit below 75 / 67 / 50 / 40 / 33 / 25 / 14 % of the starting lanes: 410 / 406 / 396 / 392 / 391 / 393 / 402 ms
                    // on cfg 3; testing only every second step: 412 ms; a decision after every step: 429 ms)
                    if ((uint32_t)__popcll(m) < stay) break;
                }
                if (COUNT) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_inner += now - t_mark; t_mark = now; }
            }
        }
    }

    {
        uint32_t r = wave_sum(n_rays);
        if (!PROBE && lane == 0 && r) atomicAdd(&P.counters[0], (unsigned long long)r);
        if (COUNT) {
            uint32_t a = wave_sum(ts.n_iters), b = wave_sum(ts.n_tri), c = wave_sum(ts.n_box);
            if (lane == 0) {
                atomicAdd(&P.counters[1], (unsigned long long)a);
                atomicAdd(&P.counters[2], (unsigned long long)b);
                atomicAdd(&P.counters[3], (unsigned long long)c);
                atomicAdd(&P.counters[4], (unsigned long long)ts.w_iters);   // wave-uniform
                atomicAdd(&P.counters[5], (unsigned long long)ts.w_alive);
            }
            const uint32_t nn = wave_sum(ts.n_nan), nh = wave_sum(n_hits);
            if (lane == 0) {
                atomicAdd(&P.counters[24], (unsigned long long)nh);
                atomicAdd(&P.counters[6], (unsigned long long)nn);
                atomicAdd(&P.counters[7], (unsigned long long)ts.w_fringe);
                atomicAdd(&P.counters[8], (unsigned long long)ts.l_fringe);
                atomicAdd(&P.counters[9], (unsigned long long)ts.l_inner);
                atomicAdd(&P.counters[10], t_shade);
            }
            // the first time any lane of the wave found the queue empty (units of 256 cycles since the wave's birth)
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            uint32_t dry_first = t_dry ? (uint32_t)((t_dry - t_born) >> 8) : 0xffffffffu;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dry_first = min(dry_first, (uint32_t)__shfl_xor((int)dry_first, off, 64));
            uint32_t wave_max_pix_rays = max_pix_rays;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wave_max_pix_rays = max(wave_max_pix_rays, (uint32_t)__shfl_xor((int)wave_max_pix_rays, off, 64));
            atomicMax(&P.counters[13], (unsigned long long)max_pix_iters);
            atomicMax(&P.counters[14], (unsigned long long)max_pix_rays);
            if (lane == 0) {
                atomicAdd(&P.counters[11], t_inner);
                atomicAdd(&P.counters[12], t_fringe);
                atomicAdd(&P.counters[15], (unsigned long long)w_shade_passes);
                atomicAdd(&P.counters[16], (unsigned long long)l_shade);
                atomicAdd(&P.counters[17], (unsigned long long)l_cam);
                atomicAdd(&P.counters[18], (unsigned long long)w_reject_iters);
                if (P.wave_debug) {
                    // per wave: end of life and of the queue (units of 256 cycles since birth), rays, the most expensive pixel (rays)
                    uint32_t *w = P.wave_debug + sizeof(OrderProfile) / 4u + 4u * (blockIdx.x * (blockDim.x >> 6) + wave);
                    w[0] = (uint32_t)((t_end - t_born) >> 8); w[1] = dry_first; w[2] = r; w[3] = wave_max_pix_rays;
                }
                atomicAdd(&P.counters[19], 1ull);
                atomicAdd(&P.counters[20], t_end - t_born);
                atomicMax(&P.counters[21], t_end - t_born);
                atomicAdd(&P.counters[22], dry_first != 0xffffffffu ? (t_end - t_born) - ((unsigned long long)dry_first << 8) : 0ull);
            }
        }
    }
}

// Pixel queue of one launch, built on the device so that srt_render_chunk never has to synchronise with the host.
//
// 1. Cost-descending order of the local tiles (longest-processing-time-first): 4096-bin counting sort on the probe's
//    per-tile cost.  The order inside a bin is arbitrary -- it only affects scheduling, never results.
// 2. Splitting.  A pixel is one sequential chain (all its samples draw from one RNG stream), so the launch cannot end
//    before its most expensive pixel does, and 64 such pixels in one wave slow each other down further (a step serves one
//    kind of work): measured on cfg 3, the most expensive tile alone takes 242 ms with 64 pixels per wave, 209 / 192 / 186 /
//    166 / 107 ms with 16 / 8 / 4 / 2 / 1.  When the launch is chain-bound -- few tiles per wave: a small chunk, or one
//    rank's share of a multi-GPU frame -- the tiles whose estimated latency exceeds the launch's estimated makespan are
//    split into 2^s rows of 64 >> s pixels, each row taken by a different wave.  With many tiles per wave (one GPU, full
//    frame) the makespan estimate is far above any tile and nothing is split.
//    queue row = tile | part << 22 | s << 28;  queue_info[0] = number of rows.
__global__ __launch_bounds__(1024) void order_tiles_kernel(const uint32_t *__restrict__ cost, uint32_t *__restrict__ sorted,
                                                          uint32_t *__restrict__ rows, uint32_t n, uint32_t n_waves,
                                                          uint32_t split_load_pct, uint32_t *__restrict__ queue_info, uint32_t order_max_pct) {
    constexpr uint32_t kBinsN = 4096;
    __shared__ uint32_t s_bin[kBinsN];
    __shared__ uint32_t s_max;
    __shared__ unsigned long long s_sum;
    __shared__ uint32_t s_scan[1024];
    __shared__ float s_load;
    const uint32_t t = threadIdx.x;
    if (t == 0) { s_max = 1u; s_sum = 0ull; }
    for (uint32_t b = t; b < kBinsN; b += 1024) s_bin[b] = 0u;
    __syncthreads();
    // Sort key of a tile: its cost (sum over its 64 pixels) moved order_max_pct % of the way towards 64 x its most expensive pixel
    // (cost[n + k], the probe's second array).  0: longest-processing-time-first by tile cost; 100: tiles holding the longest single
    // chains first, whatever the rest of the tile costs.  The split policy and the wave priorities below keep using the cost.
    auto key_of = [&](uint32_t k) -> uint32_t {
        const uint32_t c = cost[k];
        if (order_max_pct == 0u) return c;
        const unsigned long long mx64 = (unsigned long long)cost[n + k] * 64ull;
        const unsigned long long key = mx64 > c ? c + (mx64 - c) * order_max_pct / 100ull : c;
        return key > 0xffffffffull ? 0xffffffffu : (uint32_t)key;
    };
    __shared__ uint32_t s_kmax;
    if (t == 0) s_kmax = 1u;
    uint32_t m = 0, km = 0;
    unsigned long long sum = 0;
    for (uint32_t k = t; k < n; k += 1024) { m = max(m, cost[k]); km = max(km, key_of(k)); sum += cost[k]; }
    atomicMax(&s_max, m);
    atomicAdd(&s_sum, sum);
    __syncthreads();
    atomicMax(&s_kmax, km);
    __syncthreads();
    const float scale = (float)(kBinsN - 1) / (float)s_kmax;
    auto bin_of = [&](uint32_t c) { uint32_t b = (uint32_t)((float)c * scale); b = b > kBinsN - 1 ? kBinsN - 1 : b; return (kBinsN - 1) - b; };   // bin 0 = most expensive
    for (uint32_t k = t; k < n; k += 1024) atomicAdd(&s_bin[bin_of(key_of(k))], 1u);
    __syncthreads();
    if (t == 0) {       // exclusive scan of 4096 counters: trivial next to the render
        uint32_t acc = 0;
        for (uint32_t b = 0; b < kBinsN; b++) { const uint32_t c = s_bin[b]; s_bin[b] = acc; acc += c; }
    }
    __syncthreads();
    for (uint32_t k = t; k < n; k += 1024) sorted[atomicAdd(&s_bin[bin_of(key_of(k))], 1u)] = k;
    __syncthreads();

    // latency of a tile relative to the same tile at 64 pixels per wave, by split level (pixels per wave 64 .. 1)
    // (measured on the most expensive tile of cfg 3 with the v15 kernel, alone on the GPU: 164 / 153 / 152 / 131 / 123 / 108 / 80 ms for
    // 64 / 32 / 16 / 8 / 4 / 2 / 1 pixels per wave, tools/lone_tile.py; the ratios are a property of the kernel's lane sharing, not of
    // the scene: the policy only needs their order of magnitude)
    const float g[7] = {1.0f, 0.934f, 0.925f, 0.797f, 0.747f, 0.656f, 0.485f};
    auto level_for = [&](uint32_t c, float target) {
        uint32_t s = 0;
        while (s < 6u && (float)c * g[s] > target) s++;
        return s;
    };
    // Makespan target T (cost units): the smallest T such that (a) every row's latency c * g[s] fits into T with the
    // smallest possible split level s(c, T), and (b) the rows fit the machine: sum over rows of their latency -- a row
    // holds a wave slot for that long however few lanes it uses -- times a load factor <= n_waves * T.  Splitting buys
    // latency with wave-slot time (a tile cut into 64 single-pixel rows costs 28x its unsplit slot time), so (b) is what
    // keeps a throughput-bound launch from splitting anything.  Bisection; every step is one parallel reduction.
    // (also keeps 64 * rows far below 2^32: the queue head is a 32-bit pixel-slot counter)
    // (latency estimate of a tile for the split policy: its cost.  Using the sort key instead -- a tile of average cost that holds
    // one very long pixel counted like 64 such pixels -- was measured and lost everywhere: cfg 3 at W = 4 / 8 125 -> 141 / 106 ->
    // 135 ms, cfg 5 at W = 8 2245 -> 2262 ms; profiles/r04/cfg5_w8_tail.txt)
    auto lat_of = [&](uint32_t k) -> uint32_t { return cost[k]; };
    const bool may_split = split_load_pct != 0u && n <= (1u << 20);
    float target = 3.0e38f;
    if (may_split) {
        const float load_factor = (float)split_load_pct * 0.01f;
        const float lat_max = (float)s_max;
        float lo_t = lat_max * g[6], hi_t = fmaxf(lat_max, load_factor * (float)s_sum / (float)(n_waves ? n_waves : 1u));
        for (int it = 0; it < 14; it++) {
            const float mid = 0.5f * (lo_t + hi_t);
            float load = 0.f;
            for (uint32_t k = t; k < n; k += 1024) { const uint32_t c = lat_of(k); const uint32_t s = level_for(c, mid); load += (float)(1u << s) * (float)c * g[s]; }
            if (t == 0) s_load = 0.f;
            __syncthreads();
            atomicAdd(&s_load, load);
            __syncthreads();
            const bool feasible = load_factor * s_load <= (float)n_waves * mid;
            __syncthreads();
            if (feasible) hi_t = mid; else lo_t = mid;
        }
        target = hi_t;
    }
    auto level_of = [&](uint32_t k) { return may_split ? level_for(lat_of(k), target) : 0u; };
    // rows per thread over a contiguous piece of the sorted order, then a scan
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = min(n, t * per), hi = min(n, lo + per);
    uint32_t mine = 0;
    // (queue position q holds sorted[q].  Sending the head of the order -- the longest chains -- to the END of the queue, so that they run
    // when the machine has emptied, was measured on cfg 5 at W = 8 and costs 0.5 s and more: profiles/r04/cfg5_w8_tail.txt)
    for (uint32_t k = lo; k < hi; k++) mine += 1u << level_of(sorted[k]);
    s_scan[t] = mine;
    __syncthreads();
    if (t == 0) {
        uint32_t acc = 0;
        for (uint32_t b = 0; b < 1024u; b++) { const uint32_t c = s_scan[b]; s_scan[b] = acc; acc += c; }
        queue_info[0] = acc;
        queue_info[1] = s_max;      // the most expensive tile's probe cost (render_kernel's wave priorities)
    }
    __syncthreads();
    uint32_t at = s_scan[t];
    for (uint32_t k = lo; k < hi; k++) {
        const uint32_t tile = sorted[k], s = level_of(tile);
        for (uint32_t part = 0; part < (1u << s); part++) rows[at++] = tile | (part << 22) | (s << 28);
    }
}

// Gathered compact tiles -> block-linear planar framebuffer (rendering.cu:146-148 layout).
__global__ __launch_bounds__(64) void scatter_tiles_kernel(const ScatterParams P) {
    const uint32_t tile = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (tile >= P.n_tiles) return;
    const uint32_t tile_x = tile % P.tiles_x, tile_y = tile / P.tiles_x;
    const uint32_t i = tile_x * 8u + (lane & 7u), j = tile_y * 8u + (lane >> 3);
    if (i >= P.width || j >= P.height || i / P.tx >= P.bx || j / P.ty >= P.by) return;
    const uint32_t idx = block_linear_idx(i, j, P.tx, P.ty, P.bx);
    const uint32_t rank = tile % P.world, local = tile / P.world;
    for (uint32_t g = 0; g < P.groups; g++) {
        const float *src = P.gathered + (((size_t)rank * P.groups + g) * P.tiles_padded + local) * (kGroupPlanes * kTileLanes) + lane;
#pragma unroll
        for (int p = 0; p < kGroupPlanes; p++) P.fb[g * kGroupPlanes + p][idx] = src[p * kTileLanes];
    }
}

// render_manager::update_fb's un-swizzle (render_manager.cuh:68-142), one thread per block-linear index.
__global__ void unswizzle_kernel(const float *r, const float *g, const float *b, float *dr, float *dg, float *db, uint32_t tx,
                                 uint32_t ty, uint32_t bx, uint32_t by, uint32_t n_cols, uint32_t n_rows, uint32_t offx,
                                 uint32_t offy, uint32_t image_width, uint32_t image_height) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t block_size = tx * ty;
    if (idx >= block_size * bx * by) return;
    const uint32_t blk = idx / block_size, thread_idx = idx % block_size;
    const uint32_t block_x = blk % bx, block_y = blk / bx;
    uint32_t fb_x = tx * block_x + thread_idx % tx;
    uint32_t fb_y = ty * block_y + thread_idx / tx;
    if (fb_x < n_cols && fb_y < n_rows) {
        fb_x += offx; fb_y += offy;
        if (fb_x < image_width && fb_y < image_height) {
            const size_t o = (size_t)fb_y * image_width + fb_x;
            dr[o] = r[idx]; dg[o] = g[idx]; db[o] = b[idx];
        }
    }
}

// bvh::hit for explicit rays (KAT entry point)
__global__ __launch_bounds__(64) void trace_rays_kernel(const RenderParams P, const float *rays, uint32_t n, float *out) {
    extern __shared__ float4 lds4[];
    uint32_t *s_stack = reinterpret_cast<uint32_t *>(lds4);
    const uint32_t lane = threadIdx.x;
    const uint32_t k = blockIdx.x * 64u + lane;
    const bool active = k < n;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    if (active) { o = mk(rays[6 * k + 0], rays[6 * k + 1], rays[6 * k + 2]); d = mk(rays[6 * k + 3], rays[6 * k + 4], rays[6 * k + 5]); }
    TravStats ts;
    Trav tv; tv.node = kTravIdle; tv.sp = 0u; tv.top = -1; tv.c = kFltMax; tv.hit = -1; tv.nf[0] = tv.nf[1] = tv.nf[2] = 0u;
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    StackRef my_stack; my_stack.s16 = nullptr; my_stack.s32 = (lds_i32 *)s_stack + lane;
    stack_init<false>(my_stack);
    if (active) trav_begin<false>(tv, stack_base<false>(my_stack), P.tris, P.root_ref, o, d, ts);
    NodeSrc ns;
    ns.global_nodes = make_rsrc(P.nodes, (uint32_t)P.n_inner * 64u);
    ns.global_fringe = make_rsrc(P.fringe, (uint32_t)(P.n_records - P.n_inner) * P.fringe_stride);
    ns.fringe_stride = P.fringe_stride;
    ns.global_nodes_sw = make_rsrc(nullptr, 0u);
    ns.n_inner = P.n_inner;
    ns.lds_q0 = ns.lds_q1 = ns.lds_q2 = nullptr; ns.lds_r0 = ns.lds_r1 = nullptr; ns.n_cached = 0;
    while (__ballot(tv.node >= 0) != 0ull) {
        if (tv.node >= P.n_inner) trav_step_fringe<false, false>(tv, ns, o, d, inv, my_stack, ts);
        else if (tv.node >= 0) trav_step_inner<false, false, false>(tv, ns, o, inv, my_stack, ts);   // n_cached = 0: global records
    }
    const float t = tv.c;
    const int tri = tv.hit;
    if (active) {
        float ff = 0.f, mat = 0.f;
        if (tri >= 0) {
            const float4 ta = P.tris[3 * tri + 0];
            const float4 tc = P.tris[3 * tri + 2];
            ff = dot(d, mk(ta.x, ta.y, ta.z)) < 0 ? 1.f : 0.f;
            mat = (float)(__float_as_uint(tc.z) >> 8);
        }
        out[4 * k + 0] = tri >= 0 ? t : 0.f;
        out[4 * k + 1] = (float)tri;
        out[4 * k + 2] = ff;
        out[4 * k + 3] = mat;
    }
}

// Primitive-op sweep: proves device + - * / sqrt fmin fmax casts and srt_powf are bit-identical to the host.
__global__ void op_sweep_kernel(int which, const float *a, const float *b, uint32_t n, float *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float x = a[k], y = b[k];
    const float z = a[(k ^ 1u) < n ? (k ^ 1u) : k];      // third operand of kinds 20 / 21
    float r;
    switch (which) {
    case 0: r = x + y; break;
    case 1: r = x - y; break;
    case 2: r = x * y; break;
    case 3: r = x / y; break;
    case 4: r = sqrtf(x); break;
    case 5: r = fminf(x, y); break;
    case 6: r = fmaxf(x, y); break;
    case 7: r = (float)(int)x; break;
    case 8: r = dev_powf(x, y); break;
    case 9: r = 1.0f / x; break;
    case 10: r = fabsf(x); break;
    case 11: r = x * y + y; break;                      // must NOT contract into an fma
    case 12: r = (float)__float_as_uint(x); break;      // u32 -> f32 conversion (RNG uniform mapping)
    case 13: r = x * x + y * y + x * y; break;          // dot-style chain, left associated
    case 14: { f2 a2 = mk2(x, y), b2 = mk2(y, x); f2 c2 = a2 * b2; r = c2.x; if (k & 1) r = c2.y; } break;          // v_pk_mul_f32
    case 15: { f2 a2 = mk2(x, y), b2 = mk2(y, x * 0.5f); f2 c2 = a2 + b2; r = (k & 1) ? c2.y : c2.x; } break;    // v_pk_add_f32
    case 16: { f2 a2 = mk2(x, y), b2 = mk2(y, x); f2 c2 = (a2 - b2) * x - a2 * b2; r = (k & 1) ? c2.y : c2.x; } break;
    case 17: r = srt_pow5f(x); break;                  // must equal dev_powf(x, 5)
    case 18: r = rng_unit_from_bits(__float_as_uint(x)); break;   // must equal (float)r * 2^-32 + 2^-33 in two roundings
    case 19: r = rng_pm1_from_bits(__float_as_uint(x)); break;    // must equal ((float)r * 2^-32 + 2^-33) * 2 + -1
    // the inside test of the FRINGE visit (NaN-propagating minimum, sign flip for counter-clockwise triangles) on the operand
    // triple (a[k], b[k], a[k ^ 1]): must equal the three compares of is_interior_faster (tri.cu:121-128)
    case 20: { float m = inside_min2(x, y, 0u); asm volatile("" : "+v"(m)); r = inside_min3(m, z, 0u) ? 1.f : 0.f; } break;                      // a1 >= 0 && a2 >= 0 && a3 >= 0
    case 21: { float m = inside_min2(x, y, 0x80000000u); asm volatile("" : "+v"(m)); r = inside_min3(m, z, 0x80000000u) ? 1.f : 0.f; } break;    // a1 <= 0 && a2 <= 0 && a3 <= 0
    // the product's Sellmeier arithmetic with coefficients B = b[0..2], C = b[3..5] at wavelength a[k]: compared with the reference's
    // own sellmeier_index compiled from its source (tests/test_ref_tables.py)
    case 22: r = n >= 6u ? sellmeier_index(b[0], b[1], b[2], b[3], b[4], b[5], x) : 0.f; break;
    // the hand-scheduled rejection loops (srt_device.h) from the stream curand_init(seed = bits of a[k]) would start: the accepted point, its |p|^2
    // and a fold of the stream position they leave behind -- every lane of a wave needs another number of tries, so the v_cmpx bookkeeping of the
    // loops runs fully divergent here.  Checked against a host restatement of XORWOW + the C++ loops (tests/test_gpu_parity.py).
    case 23: case 24: case 25: case 26: case 27: {
        Rng s; rng_seed(s, (uint64_t)__float_as_uint(x));
        float px, py, pz, l2; rng_sphere_loop_asm(s, px, py, pz, l2);
        r = which == 23 ? px : which == 24 ? py : which == 25 ? pz : which == 26 ? l2 : __uint_as_float(s.d ^ (s.v0 * 3u) ^ (s.v1 * 5u) ^ (s.v2 * 7u) ^ (s.v3 * 11u) ^ (s.v4 * 13u));
    } break;
    case 28: case 29: case 30: {
        Rng s; rng_seed(s, (uint64_t)__float_as_uint(x));
        float px, py; rng_disk_loop_asm(s, px, py);
        r = which == 28 ? px : which == 29 ? py : __uint_as_float(s.d ^ (s.v0 * 3u) ^ (s.v1 * 5u) ^ (s.v2 * 7u) ^ (s.v3 * 11u) ^ (s.v4 * 13u));
    } break;
    // rng_next alone (the v_bitop3 form every other draw of the render kernels goes through), (bits of b[k]) steps from the stream
    // curand_init(seed = bits of a[k]) would start: xor of the outputs / weighted wrapping sum / fold of the words afterwards.  Held against
    // rocRAND's xorwow_engine::next() run from the same words (tests/test_rocrand_xorwow_pin.py).
    case 31: case 32: case 33: {
        Rng s; rng_seed(s, (uint64_t)__float_as_uint(x));
        const uint32_t steps = __float_as_uint(y);
        uint32_t xo = 0u, sm = 0u;
        for (uint32_t i = 0; i < steps; i++) { const uint32_t v = rng_next(s); xo ^= v; sm += v * (2u * i + 1u); }
        r = __uint_as_float(which == 31 ? xo : which == 32 ? sm : s.d ^ (s.v0 * 3u) ^ (s.v1 * 5u) ^ (s.v2 * 7u) ^ (s.v3 * 11u) ^ (s.v4 * 13u));
    } break;
    default: r = 0.f;
    }
    out[k] = r;
}

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------
hipError_t launch_init_rng(uint32_t *rng, uint32_t n_lanes, uint64_t seed, hipStream_t st) {
    if (n_lanes == 0) return hipSuccess;
    hipLaunchKernelGGL(init_rng_kernel, dim3((n_lanes + 255) / 256), dim3(256), 0, st, rng, n_lanes, seed);
    return hipGetLastError();
}

template <int MODE, bool NARROW, bool ALL_CACHED, bool PAIRED = false>
static hipError_t launch_render_cached(const RenderParams &p_in, const LaunchPlan &lp, const PlanKnobs &knobs, uint32_t n_cu, hipStream_t st, uint32_t *waves_launched) {
    RenderParams p = p_in;
    int wpb = lp.waves_per_block;
    uint32_t waves_per_cu = (uint32_t)lp.waves_per_cu;
    if (p.waves_per_cu_override > 0 && p.waves_per_cu_override < 16 && wpb == 16) {
        // experiment knob: fewer waves per CU (one smaller workgroup per CU, same LDS cache), e.g. 8 = two waves per SIMD
        wpb = (int)p.waves_per_cu_override;
        waves_per_cu = p.waves_per_cu_override;
    }
    p.n_cached = lp.n_cached;
    const size_t lds = render_lds_bytes(p.stack_depth, wpb, lp.n_cached, p.n_records, knobs);
    // per launch, not once per process: the attribute belongs to the function ON THE CURRENT DEVICE, and one process may
    // drive several GPUs (srt_comm_init_all); the call is a host-side table update
    {
        const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel<MODE, NARROW, ALL_CACHED, PAIRED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
        if (ae != hipSuccess) return ae;
    }
    // persistent waves: fill every CU (waves_per_cu at this build's register budget), never more waves than queue rows
    uint32_t n_waves = n_cu * waves_per_cu;
    if (n_waves > p.queue_rows_bound) n_waves = p.queue_rows_bound;     // (upper bound known to the host)
    const uint32_t n_blocks = (n_waves + (uint32_t)wpb - 1) / (uint32_t)wpb;
    if (waves_launched) *waves_launched = n_blocks * (uint32_t)wpb;
    hipLaunchKernelGGL((render_kernel<MODE, NARROW, ALL_CACHED, PAIRED>), dim3(n_blocks), dim3(64 * wpb), lds, st, p);
    return hipGetLastError();
}

template <int MODE, bool NARROW>
static hipError_t launch_render_mode(const RenderParams &p, const PlanKnobs &knobs, uint32_t n_cu, hipStream_t st, uint32_t *waves_launched) {
    LaunchPlan lp;
    render_launch_plan(p.stack_depth, p.n_records, p.n_inner, knobs, lp);
    // (the ALL_CACHED variant reads packed 96-byte FRINGE records by a literal stride: never pick it for a scene that was uploaded
    // with padded records -- the plan can only differ from the upload's when the context's test knobs changed in between)
    if (lp.all_cached && p.fringe_stride != 96u) lp.all_cached = false;
    // a paired tree (no node with one leaf child) that is LDS resident with 16-bit references runs the variant without the FRINGE box test
    if (render_paired_variant(p.paired != 0u, NARROW, lp.all_cached))
        return NARROW ? launch_render_cached<MODE, true, true, true>(p, lp, knobs, n_cu, st, waves_launched)
                      : launch_render_cached<MODE, false, false, true>(p, lp, knobs, n_cu, st, waves_launched);
    return lp.all_cached ? launch_render_cached<MODE, NARROW, true>(p, lp, knobs, n_cu, st, waves_launched) : launch_render_cached<MODE, NARROW, false>(p, lp, knobs, n_cu, st, waves_launched);
}

hipError_t launch_render(const RenderParams &p, const PlanKnobs &knobs, uint32_t n_cu, int mode, hipStream_t st, uint32_t *waves_launched) {
    if (waves_launched) *waves_launched = 0;
    if (p.tiles_local == 0) return hipSuccess;
    const bool narrow = render_narrow_refs(p.n_records, knobs);
    if (mode == 1) return narrow ? launch_render_mode<1, true>(p, knobs, n_cu, st, waves_launched) : launch_render_mode<1, false>(p, knobs, n_cu, st, waves_launched);
    if (mode == 2) return narrow ? launch_render_mode<2, true>(p, knobs, n_cu, st, waves_launched) : launch_render_mode<2, false>(p, knobs, n_cu, st, waves_launched);
    return narrow ? launch_render_mode<0, true>(p, knobs, n_cu, st, waves_launched) : launch_render_mode<0, false>(p, knobs, n_cu, st, waves_launched);
}

hipError_t launch_order_tiles(const uint32_t *cost, uint32_t *sorted, uint32_t *rows, uint32_t n, uint32_t n_waves,
                              uint32_t split_load_pct, uint32_t *queue_info, uint32_t order_max_pct, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(order_tiles_kernel, dim3(1), dim3(1024), 0, st, cost, sorted, rows, n, n_waves, split_load_pct, queue_info, order_max_pct);
    return hipGetLastError();
}

hipError_t launch_scatter(const ScatterParams &p, hipStream_t st) {
    if (p.n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_tiles_kernel, dim3(p.n_tiles), dim3(64), 0, st, p);
    return hipGetLastError();
}

hipError_t launch_unswizzle(const float *const src[3], float *const dst[3], uint32_t tx, uint32_t ty, uint32_t bx, uint32_t by,
                            uint32_t n_cols, uint32_t n_rows, uint32_t offx, uint32_t offy, uint32_t image_width,
                            uint32_t image_height, hipStream_t st) {
    const uint32_t n = tx * ty * bx * by;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unswizzle_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src[0], src[1], src[2], dst[0], dst[1], dst[2],
                       tx, ty, bx, by, n_cols, n_rows, offx, offy, image_width, image_height);
    return hipGetLastError();
}

hipError_t launch_trace(const RenderParams &p, const float *rays, size_t n, float *out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const size_t lds = stack_slots(p.stack_depth) * 64 * 4;
    hipLaunchKernelGGL(trace_rays_kernel, dim3((uint32_t)((n + 63) / 64)), dim3(64), lds, st, p, rays, (uint32_t)n, out);
    return hipGetLastError();
}

hipError_t launch_op_sweep(int which, const float *a, const float *b, size_t n, float *out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(op_sweep_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, which, a, b, (uint32_t)n, out);
    return hipGetLastError();
}

}  // namespace srt
