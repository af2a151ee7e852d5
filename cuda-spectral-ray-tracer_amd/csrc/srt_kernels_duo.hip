// srt_kernels_duo.hip -- the two-context render kernel: the same hot path as render_kernel (srt_kernels.hip;
// spectral_render_kernel, rendering/rendering.cu:151-235), regrouped so that a lane is never idle while one of its pixels waits.
//
// render_kernel's lane owns ONE pixel; its lane fill is 27 of 64 per vector instruction for two reasons the counters name
// (profiles/r03): during a traversal step ~18 lanes wait for a shading pass, and the shading pass itself runs its branches --
// scatter by material, path end + next camera ray -- one after the other over whichever lanes need them (22 of 64 lanes per
// instruction).  Deferring lanes to batch them was measured slower every time: a waiting lane costs more than the instructions
// it saves.  Here a lane owns TWO pixels (contexts):
//   context A  the ray the traversal steps walk (origin, direction, reciprocal, closest t, hit) + its pixel's state
//   context S  the other pixel: waiting for a shading pass, or holding its next ray ("pending")
// Traversal code only touches A, shading code only touches S; when A's ray has finished and S holds a pending ray, a SWAP step
// exchanges the two contexts (35 registers) and the lane walks on at once.  A lane idles only when both its pixels wait for
// shading -- and since waiting no longer idles a lane, shading is batched BY CLASS, each class a pass of its own that runs when
// enough lanes wait for it (or when lanes are blocked):
//   pass D  hit on a lambertian / metallic / default material: random_unit_vector + scatter           (material.cu:8-37,64-71,88-92)
//   pass G  hit on a dielectric: Sellmeier index, Schlick, reflect / refract                           (material.cu:73-80,102-136)
//   pass E  path end -- miss (background), emissive hit, absorbed / exhausted path: spectrum -> XYZ, then the pixel switch
//           and the next camera ray                                                      (rendering.cu:24-27,38,66-87,140-149,215-232)
// The class of a hit costs nothing to find: the FRINGE records carry it in bits 28-30 of a leaf's child reference (flatten_scene),
// so it arrives in the hit word.  Every pixel still consumes its own XORWOW stream in the reference's order and every arithmetic
// expression is the one render_kernel evaluates: the image is bit-identical (the GPU suite holds both kernels to the same CPU oracle).
//
// Used for launches that are throughput-bound (many tiles per wave: the 1-GPU frame): a pixel's chain advances at half the speed
// when it shares its lane, so chain-bound launches (a rank's share of an 8-GPU frame, small chunks) keep render_kernel.
// Production build only (no counters, 15-bit record references, whole inner tree in LDS); tiles are never split.
#include <algorithm>
#include <stdlib.h>

#include "srt_color_consts.h"
#include "srt_kernel_common.h"

namespace srt {

namespace {

constexpr uint32_t kStEmpty = 0u;        // no pixel: fetch one (pass E)
constexpr uint32_t kStNeedSample = 1u;   // pixel, no path: next camera ray, or pixel end when all samples are drawn (pass E)
constexpr uint32_t kStResult = 2u;       // closest-hit query finished: (c, hit) wait for the pass of their class
constexpr uint32_t kStRay = 3u;          // holds a ray: pending in S, under way (or just finished) in A
constexpr uint32_t kStDead = 4u;         // pixel queue ran dry for this context
constexpr uint32_t kHitTriMask = 0x0fffffffu;

}  // namespace

__global__ __launch_bounds__(768) void render_kernel_duo(const RenderParams P) {
    extern __shared__ float4 lds4[];
    lds_uniforms *U = (lds_uniforms *)lds4;
    float4 *s_cmf = lds4 + kLdsUniF4;
    const uint32_t nc = (uint32_t)P.n_cached;
    float4 *s_q0 = lds4 + kLdsTablesF4, *s_q1 = s_q0 + nc, *s_q2 = s_q1 + nc;
    uint32_t *s_r0 = reinterpret_cast<uint32_t *>(s_q2 + nc);
    const size_t cache_b = (((size_t)nc * 52u) + 15u) & ~(size_t)15u;
    char *s_stack_base = reinterpret_cast<char *>(lds4 + kLdsTablesF4) + cache_b;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;

    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; a++) {
            U->du[a] = P.du[a]; U->dv[a] = P.dv[a]; U->p00[a] = P.p00[a]; U->center[a] = P.center[a];
            U->disk_u[a] = P.disk_u[a]; U->disk_v[a] = P.disk_v[a];
        }
        U->defocus_angle = P.defocus_angle;
        U->n_rows = P.queue_rows ? P.queue_rows[0] : P.tiles_local;
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
        s_r0[k] = (__float_as_uint(q3.x) & 0xffffu) | (__float_as_uint(q3.y) << 16);
    }
    __syncthreads();      // the only barrier

    NodeSrc ns;
    ns.global_nodes = make_rsrc(P.nodes, (uint32_t)P.n_inner * 64u);
    ns.fringe_stride = 96u;
    ns.global_fringe = make_rsrc(P.fringe, (uint32_t)(P.n_records - P.n_inner) * 96u);
    ns.n_inner = P.n_inner;
    ns.lds_q0 = (lds_cf4 *)s_q0; ns.lds_q1 = (lds_cf4 *)s_q1; ns.lds_q2 = (lds_cf4 *)s_q2;
    ns.lds_r0 = (lds_cu32 *)s_r0; ns.lds_r1 = (lds_cu32 *)s_r0;
    ns.n_cached = P.n_cached;
    const buf_rsrc shade_rsrc = make_rsrc(P.shade, P.n_tris * 48u);
    const buf_rsrc sd_rsrc = make_rsrc(P.mat_sd, (P.n_materials + 1u) * 768u);
    const uint32_t spp = P.spp;
    uint32_t w_swap = P.duo_w_swap, w_blk = P.duo_w_blocked, w_fringe = P.score_fringe;
    uint32_t t_d = P.duo_fill_d, t_g = P.duo_fill_g, t_e = P.duo_fill_e;
    asm volatile("" : "+s"(w_swap), "+s"(w_blk), "+s"(w_fringe), "+s"(t_d), "+s"(t_g), "+s"(t_e));
    StackRef my_stack;
    {
        const size_t slots = (size_t)(P.stack_depth < 1 ? 1 : P.stack_depth) + kStackSentinels;
        my_stack.s16 = (lds_i16 *)(s_stack_base + (size_t)wave * slots * 128u) + lane;
        my_stack.s32 = nullptr;
        stack_init<true>(my_stack);
    }
    const uint32_t n_inner_u = (uint32_t)P.n_inner;
    const float delta_lambda = (kLambdaMax - kLambdaMin) / (float)kWavelengths;

    // ---- context S: the pixel the shading passes work on ------------------------------------------------------------------
    uint32_t st_s = kStEmpty;
    Rng rs; rs.d = rs.v0 = rs.v1 = rs.v2 = rs.v3 = rs.v4 = 0u;
    V3 acc = mk(0.f, 0.f, 0.f);
    float hero = kLambdaMin;
    float pw[kWavelengths];
#pragma unroll
    for (int k = 0; k < kWavelengths; k++) pw[k] = 0.f;
    uint32_t idx = 0, out_slot = 0, pixel_ij = 0, sample = 0, bounce = 0, valid = 0;
    V3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 1.f), inv = mk(0.f, 0.f, 1.f);
    float c_s = kFltMax;
    int hit_s = -1;
    // ---- context A: the pixel whose ray the traversal steps walk ----------------------------------------------------------
    uint32_t st_a = kStEmpty;
    Rng rs_a; rs_a.d = rs_a.v0 = rs_a.v1 = rs_a.v2 = rs_a.v3 = rs_a.v4 = 0u;
    V3 acc_a = mk(0.f, 0.f, 0.f);
    float hero_a = kLambdaMin;
    float pw_a[kWavelengths];
#pragma unroll
    for (int k = 0; k < kWavelengths; k++) pw_a[k] = 0.f;
    uint32_t idx_a = 0, out_slot_a = 0, pixel_ij_a = 0, sample_a = 0, bounce_a = 0, valid_a = 0;
    V3 ro_a = mk(0.f, 0.f, 0.f), rd_a = mk(0.f, 0.f, 1.f), inv_a = mk(0.f, 0.f, 1.f);
    Trav tv; tv.node = kTravIdle; tv.sp = 0u; tv.top = -1; tv.c = kFltMax; tv.hit = -1; tv.nf[0] = tv.nf[1] = tv.nf[2] = 0u;
    TravStats ts;
    uint32_t n_rays = 0;
#ifdef SRT_DUO_STATS
    // experiment build (tools/build_variant.sh duostats -DSRT_DUO_STATS): wave-level event counts, read back through srt_get_stats
    uint32_t q_iter = 0, q_swap = 0, q_swap_l = 0, q_d = 0, q_d_l = 0, q_g = 0, q_g_l = 0, q_e = 0, q_e_l = 0, q_blk = 0, q_fr = 0, q_fr_l = 0, q_asm = 0, q_trav_l = 0, q_cam_l = 0;
#define SRT_Q(x) x
#else
#define SRT_Q(x)
#endif

    for (;;) {
        SRT_Q(q_iter++;)
        // =========================== service phase ==================================================================
        // Which S contexts wait for which pass.  A lane is BLOCKED when its A side has nothing to walk and its S side must be
        // shaded before it can take over.
        const bool a_idle = tv.node < 0;
        const bool s_res = st_s == kStResult;
        const uint32_t cls = ((uint32_t)hit_s >> 28) & 7u;                     // material class of the hit (flatten_scene); a miss has hit < 0
        const bool want_d = s_res && hit_s >= 0 && cls <= 1u;
        const bool want_g = s_res && hit_s >= 0 && cls == 2u;
        const bool want_e = (s_res && (hit_s < 0 || cls >= 3u)) || st_s == kStEmpty || st_s == kStNeedSample;
        const bool s_ready = st_s == kStRay || st_s == kStDead;                // S can take over (or has nothing to wait for)
        const bool lane_dead = a_idle && st_a != kStRay && st_s == kStDead;
        const bool blocked = a_idle && !s_ready;
        const unsigned long long m_d = __ballot(want_d), m_g = __ballot(want_g), m_e = __ballot(want_e);
        const unsigned long long m_blocked = __ballot(blocked);
        const unsigned long long m_trav = __ballot(!a_idle);
        if ((m_d | m_g | m_e) != 0ull) {
            // blocked lanes are urgent when they outweigh what the traversing lanes could do meanwhile (always when nothing walks)
            const unsigned long long m_fr = __ballot(tv.node >= (int)n_inner_u);
            const uint32_t n_fr = (uint32_t)__popcll(m_fr), n_in = (uint32_t)__popcll(m_trav) - n_fr;
            const uint32_t trav_score = max(n_fr * w_fringe, n_in << 8);
            const bool urgent = (uint32_t)__popcll(m_blocked) * w_blk > trav_score || m_trav == 0ull;
            SRT_Q(q_blk += (uint32_t)__popcll(m_blocked);)
            const bool run_d = (uint32_t)__popcll(m_d) >= t_d || (urgent && (m_d & m_blocked) != 0ull);
            const bool run_g = (uint32_t)__popcll(m_g) >= t_g || (urgent && (m_g & m_blocked) != 0ull);
            const bool run_e = (uint32_t)__popcll(m_e) >= t_e || (urgent && (m_e & m_blocked) != 0ull);

            // ---- passes D and G: one iteration of ray_bounce's loop for a hit (rendering.cu:22-36) -------------------------
            // (one body for both, entered per class: the common head and tail are the same instructions, the material branch in
            // the middle is taken by one class per pass)
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {
                const bool go = pass == 0 ? (run_d && want_d) : (run_g && want_g);
                if (__ballot(go) == 0ull) continue;
                SRT_Q(if (pass == 0) { q_d++; q_d_l += (uint32_t)__popcll(__ballot(go)); } else { q_g++; q_g_l += (uint32_t)__popcll(__ballot(go)); })
                if (go) {
                    float wl[kWavelengths];
                    hero_expand(hero, wl);
                    // rebuild the hit record from (t, triangle): tri::hit tail (tri.cu:36-39) + set_face_normal
                    const uint32_t srec = __umul24((uint32_t)hit_s, 48u);      // (24-bit multiply: the class bits above bit 24 drop out)
                    const f4v s0 = buf_load16(shade_rsrc, srec), s1 = buf_load16(shade_rsrc, srec + 16u);
                    const V3 n_geo = mk(s0.x, s0.y, s0.z);
                    const V3 hp = ro + c_s * rd;                                           // ray::at, ray.cuh:31-34
                    const bool front_face = dot(rd, n_geo) < 0;                            // hit_record.cuh:41
                    const V3 n = front_face ? n_geo : -n_geo;
                    const uint32_t mat = __float_as_uint(s0.w);
                    const uint32_t mtype = __float_as_uint(s1.x);
                    const float fuzz = s1.y;
                    // material::scatter (materials/material.cu:55-100)
                    V3 scatter_direction = mk(0.f, 0.f, 0.f);
                    float eps_sign = 1.0f;
                    bool did_scatter = true;
                    if (pass == 1) {                                                        // DIELECTRIC, :73-80
                        const f4v s2 = buf_load16(shade_rsrc, srec + 32u);
                        const V3 unit_in = unit_vector(rd);
                        float ir = sellmeier_index(s1.z, s1.w, s2.x, s2.y, s2.z, s2.w, wl[0]);
                        float refraction_ratio = front_face ? (1.0f / ir) : ir;             // refraction_scatter, :102-136
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
                        const V3 ruv = unit_vector(random_in_unit_sphere(rs));             // vec3.cuh:221-227
                        if (mtype == 1u) {
                            const V3 unit_in = unit_vector(rd);
                            V3 reflected = reflect(unit_in, n);                            // reflection_scatter, :22-37
                            scatter_direction = reflected + fuzz * ruv;
                            did_scatter = dot(scatter_direction, n) > 0;
                            if (!did_scatter) valid = 0;
                        } else {
                            scatter_direction = n + ruv;                                   // lambertian_scatter, :8-19
                            if (near_zero(scatter_direction)) scatter_direction = n;
                        }
                    }
                    ro = hp + (eps_sign * kEpsilon) * n;                                    // :96 (Q9)
                    rd = scatter_direction;                                                 // :97
                    // r_in.mul_spectrum(spectral_distribution) (:95), all seven look-ups together (see render_kernel)
                    const uint32_t sd_base = __umul24(mat, 768u);
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
                    bool begin_trav = false;
                    if (did_scatter) {
                        bounce++;
                        if (bounce < P.bounce_limit) begin_trav = true;
                        else valid = 0;                                                     // loop exhausted, :38 (Q7)
                    }
                    // A path that ends here -- absorbed by the metal (valid = 0, :69-70) or out of bounces (valid = 0) -- adds
                    // dev_spectrum_to_XYZ(valid = 0) = (0, 0, 0) to pixel_color: x + (+0) = x for every value the sum can hold (it
                    // is a sum of products of non-negative factors, never -0), so nothing is added.
                    st_s = kStNeedSample;
                    if (begin_trav) {
                        // start of the closest-hit query: bvh::hit(r, 0, FLT_MAX, rec, root) (rendering.cu:24)
                        n_rays++;
                        inv = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);                    // aabb.cu:17, hoisted out of the box test
                        c_s = kFltMax; hit_s = -1;
                        // a direction with a NaN component hits nothing (trav_begin, srt_device.h): the result is known
                        st_s = (rd.x != rd.x || rd.y != rd.y || rd.z != rd.z) ? kStResult : kStRay;
                    }
                }
            }

            // ---- pass E: path end, pixel switch, next camera ray ----------------------------------------------------------
            if (run_e && __ballot(want_e) != 0ull) {
                bool switched = false;
                SRT_Q(q_e++; q_e_l += (uint32_t)__popcll(__ballot(want_e));)
                if (want_e) {
                    // E1: a miss multiplies the background into the path (rendering.cu:24-27), an emissive hit its own spectrum
                    // (material.cu:83-86,95); then pixel_color += dev_spectrum_to_XYZ(...) (rendering.cu:227, color.cu:88-104)
                    if (st_s == kStResult) {
                        float wl[kWavelengths];
                        hero_expand(hero, wl);
                        uint32_t sd_table = P.n_materials;
                        if (hit_s >= 0) sd_table = __float_as_uint(buf_load16(shade_rsrc, __umul24((uint32_t)hit_s, 48u)).w);
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
                        float xyz_x = 0.0f, xyz_y = 0.0f, xyz_z = 0.0f;
#pragma unroll
                        for (int k = 0; k < kWavelengths; k++) {
                            const float4 r0 = s_cmf[off[k]], r1 = s_cmf[off[k] + 1];
                            const float w = wgt[k], power = pw[k];
                            const float tx_ = ((1.0f - w) * r0.x + w * r1.x) * power * delta_lambda;
                            const float ty_ = ((1.0f - w) * r0.y + w * r1.y) * power * delta_lambda;
                            const float tz_ = ((1.0f - w) * r0.z + w * r1.z) * power * delta_lambda;
                            // the reference sums the first `valid` terms; the others are replaced by +0, which leaves a sum unchanged
                            const bool live = (uint32_t)k < valid;
                            xyz_x += live ? tx_ : 0.0f;
                            xyz_y += live ? ty_ : 0.0f;
                            xyz_z += live ? tz_ : 0.0f;
                        }
                        acc = acc + mk(xyz_x, xyz_y, xyz_z);
                        st_s = kStNeedSample;
                    }
                    // E2: pixel switch: all samples of the pixel drawn (or no pixel yet)
                    if (st_s == kStEmpty || sample == spp) {
                        switched = true;
                        if (st_s == kStNeedSample) {
                            // store RNG state (rendering.cu:232) and save_to_fb (rendering.cu:140-149)
                            {
                                uint32_t *rng = join_ptr<uint32_t>(U->rng[0], U->rng[1]);
                                const size_t nl = U->n_lanes;
                                rng[0 * nl + idx] = rs.d; rng[1 * nl + idx] = rs.v0; rng[2 * nl + idx] = rs.v1;
                                rng[3 * nl + idx] = rs.v2; rng[4 * nl + idx] = rs.v3; rng[5 * nl + idx] = rs.v4;
                            }
                            const float inv_spp = 1.0f / (float)spp;
                            const V3 c = inv_spp * acc;
                            const float r_lin = (SRT_XYZ2RGB_00 * c.x) + (SRT_XYZ2RGB_01 * c.y) + (SRT_XYZ2RGB_02 * c.z);
                            const float g_lin = (SRT_XYZ2RGB_10 * c.x) + (SRT_XYZ2RGB_11 * c.y) + (SRT_XYZ2RGB_12 * c.z);
                            const float b_lin = (SRT_XYZ2RGB_20 * c.x) + (SRT_XYZ2RGB_21 * c.y) + (SRT_XYZ2RGB_22 * c.z);
                            const float r = correct_channel(r_lin), g = correct_channel(g_lin), b = correct_channel(b_lin);
                            float *o = join_ptr<float>(U->tile_out[0], U->tile_out[1]) + out_slot;
                            const size_t gs = U->tile_group_stride;
                            o[0 * kTileLanes] = (float)(int)(r * 255.99f);      // expand_sRGB (color.cu:43-49, Q15)
                            o[1 * kTileLanes] = (float)(int)(g * 255.99f);
                            o[2 * kTileLanes] = (float)(int)(b * 255.99f);
                            if (P.write_parity) {
                                o[gs + 0 * kTileLanes] = r; o[gs + 1 * kTileLanes] = g; o[gs + 2 * kTileLanes] = b;
                                o[2 * gs + 0 * kTileLanes] = acc.x; o[2 * gs + 1 * kTileLanes] = acc.y; o[2 * gs + 2 * kTileLanes] = acc.z;
                            }
                        }
                        st_s = kStEmpty;
                        // fetch the next pixel of this rank's queue (wave-aggregated atomic); skip slots outside the chunk
                        bool searching = true;
                        while (searching) {
                            const unsigned long long m = __ballot(1);
                            const int leader = __ffsll((long long)m) - 1;
                            uint32_t base = 0;
                            // the first row of a wave is assigned (spread over the CUs by cost band), later ones come from the counter: see render_kernel
                            const bool assigned = SRT_ASSIGN_FIRST_ROW != 0 && m == ~0ull && ((U->first_row_taken >> wave) & 1u) == 0u;
                            if (assigned) {
                                if (lane == 0) atomicOr((unsigned int *)&U->first_row_taken, 1u << wave);
                                base = (wave * gridDim.x + blockIdx.x) * 64u;
                            }
                            else {
                                if ((int)lane == leader) base = atomicAdd(join_ptr<uint32_t>(U->pixel_counter[0], U->pixel_counter[1]), (uint32_t)__popcll(m));
                                base = (uint32_t)__shfl((int)base, leader, 64) + (SRT_ASSIGN_FIRST_ROW != 0 ? U->n_assigned_slots : 0u);
                            }
                            const uint32_t pix = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                            if (pix >= U->n_rows * 64u) { st_s = kStDead; searching = false; }
                            else {
                                const uint32_t *tile_order = join_ptr<const uint32_t>(U->tile_order[0], U->tile_order[1]);
                                const uint32_t tile_local = tile_order ? (tile_order[pix >> 6] & 0x3fffffu) : (pix >> 6);      // (rows are whole tiles here)
                                const uint32_t lt = pix & 63u;
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
                                    st_s = kStNeedSample;
                                    searching = false;
                                }
                            }
                        }
                        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0): drain this block's traffic once per pixel (see render_kernel)
                    }
                }
#if SRT_PRIO_MODE
                // wave priority, least slack first (see render_kernel): the longest chain either context of any lane still holds
                if (__ballot(switched) != 0ull) {
                    float rem = 0.f;
                    const uint32_t *tc = join_ptr<const uint32_t>(U->prio_cost[0], U->prio_cost[1]);
                    if (tc) {
                        if (st_s != kStEmpty && st_s != kStDead) rem = (float)tc[out_slot / (uint32_t)(kGroupPlanes * kTileLanes)] * (float)(spp - sample);
                        if (st_a == kStRay) rem = fmaxf(rem, (float)tc[out_slot_a / (uint32_t)(kGroupPlanes * kTileLanes)] * (float)(spp - sample_a));
                    }
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
                // E3: new camera ray: renderer::get_ray (rendering.cu:66-87)
                SRT_Q(q_cam_l += (uint32_t)__popcll(__ballot(want_e && st_s == kStNeedSample && sample < spp));)
                if (want_e && st_s == kStNeedSample && sample < spp) {
                    float px = -0.5f + rng_uniform(rs);                       // pixel_sample_square, :49-56
                    float py = -0.5f + rng_uniform(rs);
                    const V3 du = mk(U->du[0], U->du[1], U->du[2]), dv = mk(U->dv[0], U->dv[1], U->dv[2]);
                    const V3 cam_center = mk(U->center[0], U->center[1], U->center[2]);
                    const V3 pixel_center = (mk(U->p00[0], U->p00[1], U->p00[2]) + (float)(U->offx + (pixel_ij & 0xffffu)) * du) + (float)(U->offy + (pixel_ij >> 16)) * dv;
                    V3 pixel_sample = pixel_center + (px * du + py * dv);
                    V3 origin = cam_center;
                    if (!(U->defocus_angle <= 0.0f)) {                          // defocus_disk_sample, :42-47
                        float dx, dy;
                        for (;;) {                                             // random_in_unit_disk, vec3.cuh:240-246
                            dx = rng_pm1(rs);
                            dy = rng_pm1(rs);
                            if ((dx * dx + dy * dy) + 0.0f * 0.0f < 1.0f) break;
                        }
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
                    if (P.bounce_limit > 0) {
                        n_rays++;
                        inv = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        c_s = kFltMax; hit_s = -1;
                        st_s = (rd.x != rd.x || rd.y != rd.y || rd.z != rd.z) ? kStResult : kStRay;
                    } else {
                        valid = 0;      // ray_bounce's loop body never runs (rendering.cu:22,38): contributes dev_spectrum_to_XYZ(valid = 0) = 0
                    }
                }
            }
        }

        // ---- swap step: A has nothing to walk and S is ready to take over -------------------------------------------------------
        {
            const bool do_swap = tv.node < 0 && (st_s == kStRay || (st_s == kStDead && st_a == kStRay));
            if (__ballot(do_swap) != 0ull) {
                SRT_Q(q_swap++; q_swap_l += (uint32_t)__popcll(__ballot(do_swap));)
                if (do_swap) {
// (v_swap_b32 by hand: left to the compiler, the exchange became ~180 v_mov through temporaries)
#define SRT_SWAP(a, b) asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                    SRT_SWAP(rs.d, rs_a.d) SRT_SWAP(rs.v0, rs_a.v0) SRT_SWAP(rs.v1, rs_a.v1) SRT_SWAP(rs.v2, rs_a.v2) SRT_SWAP(rs.v3, rs_a.v3) SRT_SWAP(rs.v4, rs_a.v4)
                    SRT_SWAP(acc.x, acc_a.x) SRT_SWAP(acc.y, acc_a.y) SRT_SWAP(acc.z, acc_a.z)
                    SRT_SWAP(hero, hero_a)
#pragma unroll
                    for (int k = 0; k < kWavelengths; k++) SRT_SWAP(pw[k], pw_a[k])
                    SRT_SWAP(idx, idx_a) SRT_SWAP(out_slot, out_slot_a) SRT_SWAP(pixel_ij, pixel_ij_a)
                    SRT_SWAP(sample, sample_a) SRT_SWAP(bounce, bounce_a) SRT_SWAP(valid, valid_a)
                    SRT_SWAP(ro.x, ro_a.x) SRT_SWAP(ro.y, ro_a.y) SRT_SWAP(ro.z, ro_a.z)
                    SRT_SWAP(rd.x, rd_a.x) SRT_SWAP(rd.y, rd_a.y) SRT_SWAP(rd.z, rd_a.z)
                    SRT_SWAP(inv.x, inv_a.x) SRT_SWAP(inv.y, inv_a.y) SRT_SWAP(inv.z, inv_a.z)
                    SRT_SWAP(c_s, tv.c) SRT_SWAP(hit_s, tv.hit)
                    SRT_SWAP(st_s, st_a)
#undef SRT_SWAP
                    // what was under way in A is a finished query now; what was pending in S starts
                    if (st_s == kStRay) st_s = kStResult;
                    if (st_a == kStRay) {
                        ray_near_addresses(ns, inv_a, tv.nf);
                        tv.node = P.root_ref; tv.sp = stack_base<true>(my_stack); tv.top = -1;      // trav_begin (bvh.cu:101-119); c, hit came with the context
                    } else {
                        tv.node = kTravIdle;
                    }
                }
            }
        }

        // =========================== traversal phase =========================================================
        {
            const bool a_idle2 = tv.node < 0;
            const bool dead2 = a_idle2 && st_a != kStRay && st_s == kStDead;
            const unsigned long long alive = __ballot(!dead2);
            if (alive == 0ull) break;
            const unsigned long long spend = __ballot(st_s == kStRay || st_s == kStDead);
            SRT_Q(q_asm++; q_trav_l += (uint32_t)__popcll(__ballot(tv.node >= 0));)
            while (inner_phase_duo_asm(tv, ns, ro_a, inv_a, n_inner_u, alive, spend, w_swap, w_blk, w_fringe) != 0u) {
                SRT_Q(q_fr++; q_fr_l += (uint32_t)__popcll(__ballot(tv.node >= (int)n_inner_u));)
                if (tv.node >= (int)n_inner_u) trav_step_fringe<false, true>(tv, ns, ro_a, rd_a, inv_a, my_stack, ts);
            }
        }
    }

    {
        uint32_t r = wave_sum(n_rays);
        if (lane == 0 && r) atomicAdd(&P.counters[0], (unsigned long long)r);
#ifdef SRT_DUO_STATS
        if (lane == 0) {
            const uint32_t q[15] = {q_iter, q_swap, q_swap_l, q_d, q_d_l, q_g, q_g_l, q_e, q_e_l, q_blk, q_fr, q_fr_l, q_asm, q_trav_l, q_cam_l};
            for (int k = 0; k < 15; k++) atomicAdd(&P.counters[1 + k], (unsigned long long)q[k]);
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------------
size_t render_duo_lds_bytes(int stack_depth, int n_inner) {
    const size_t slots = (size_t)(stack_depth < 1 ? 1 : stack_depth) + kStackSentinels;
    return (size_t)kLdsTablesF4 * 16 + ((((size_t)n_inner * 52u) + 15u) & ~(size_t)15u) + (size_t)kDuoWavesPerBlock * slots * 128u;
}

// the scene qualifies when the classic launch plan keeps the whole inner tree in LDS with 15-bit references and the root is a record
bool render_duo_eligible(int stack_depth, int n_records, int n_inner, int root_ref) {
    if (root_ref < 0 || n_records > 32767) return false;
    LaunchPlan lp;
    render_launch_plan(stack_depth, n_records, n_inner, lp);
    return lp.all_cached && lp.waves_per_block == 16 && render_duo_lds_bytes(stack_depth, n_inner) <= kLdsBudget;
}

hipError_t launch_render_duo(const RenderParams &p_in, uint32_t n_cu, hipStream_t st) {
    if (p_in.tiles_local == 0) return hipSuccess;
    RenderParams p = p_in;
    p.n_cached = p.n_inner;
    const size_t lds = render_duo_lds_bytes(p.stack_depth, p.n_inner);
    {
        const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel_duo), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
        if (ae != hipSuccess) return ae;
    }
    uint32_t n_waves = n_cu * (uint32_t)kDuoWavesPerBlock;
    if (n_waves > p.queue_rows_bound) n_waves = p.queue_rows_bound;
    const uint32_t n_blocks = (n_waves + (uint32_t)kDuoWavesPerBlock - 1) / (uint32_t)kDuoWavesPerBlock;
    hipLaunchKernelGGL(render_kernel_duo, dim3(n_blocks), dim3(64 * kDuoWavesPerBlock), lds, st, p);
    return hipGetLastError();
}

}  // namespace srt
