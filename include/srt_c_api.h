/*
 * srt_c_api.h -- C-ABI of the MI355X-native spectral path-tracing hot path.
 *
 * This is the drop-in boundary for PieSil/CUDA-spectral-ray-tracer's render path.  The reference has
 * no FFI layer: the path sits behind the host classes `renderer` (rendering/rendering.cuh:39-155) and
 * `render_manager` (rendering/render_manager.cuh:37-224) and consumes device-heap objects built by
 * `scene_manager` (scene/scene.cuh:103-176).  Each entry point below names the reference interface it
 * replaces.  Plain pointers and sizes only; no C++/torch types.  All functions return 0 on success and
 * a negative srt_status on failure (never exit(): the reference's checkCudaErrors -> exit(99),
 * utils/cuda_utility.cu:8-18, is deliberately not reproduced); srt_last_error() gives the message.
 *
 * Host-side objects (srt_scene) need no GPU.  A device context (srt_ctx) needs a gfx950 device and
 * fails loudly without one -- there is no CPU fallback behind this API.
 */
#ifndef SRT_C_API_H
#define SRT_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRT_API __attribute__((visibility("default")))

#define SRT_N_CIE_SAMPLES 95      /* utils/cie_const.cuh:8 */
#define SRT_N_RAY_WAVELENGTHS 7   /* ray/ray.cuh:12 */
#define SRT_DEFAULT_SEED 1984u    /* rendering/rendering.cu:137, scene/scene.cu:14 */
#define SRT_DEFAULT_TX 28u        /* rendering/render_manager.cu:93-94 */
#define SRT_DEFAULT_TY 16u

typedef enum {
    SRT_OK = 0,
    SRT_ERR_INVALID = -1,      /* bad argument / call order (reference: message on cerr, call ignored) */
    SRT_ERR_NO_DEVICE = -2,    /* no gfx950 device / HIP runtime error at context creation */
    SRT_ERR_HIP = -3,          /* HIP runtime error (reference: checkCudaErrors -> exit(99)) */
    SRT_ERR_BVH = -4,          /* BVH build failed (reference: scene.cu:413-416 "Error building BVH") */
    SRT_ERR_UNSUPPORTED = -5,  /* e.g. non-grey sRGB colour without the (missing) rgb2spec table */
    SRT_ERR_NOMEM = -6
} srt_status;

/* material_type ids, materials/material.cuh:16-22 */
enum { SRT_MAT_LAMBERTIAN = 0, SRT_MAT_METALLIC = 1, SRT_MAT_DIELECTRIC = 2, SRT_MAT_EMISSIVE = 4, SRT_MAT_NO_MAT = 6 };
/* AAPlane, primitives/tri.cuh:8-13 */
enum { SRT_AAP_NONE = 0, SRT_AAP_XY = 1, SRT_AAP_YZ = 2, SRT_AAP_XZ = 3 };
/* scene ids: the reference's three (io/params.h:15-19) plus this build's synthetic benchmark scenes */
enum { SRT_SCENE_CORNELL = 0, SRT_SCENE_PRISM = 1, SRT_SCENE_TRIS = 2, SRT_SCENE_RANDOM_SPHERES = 100, SRT_SCENE_MESH100K = 101 };
/* BVH builders */
enum { SRT_BVH_REFERENCE = 0,  /* bit-faithful reference topology, bvh/bvh.cu:206-346 (x/y-only median split, Q14) */
       SRT_BVH_SAH = 1 };      /* this build's binned-SAH builder (same node semantics, better tree) */

/* Raw triangle as scene construction leaves it, before tri::init (primitives/tri.cu:47-84).
 * aa_plane is the value the member holds BEFORE init runs (sticky); NONE for a fresh tri. */
typedef struct { float v0[3], v1[3], v2[3]; uint32_t mat_index; uint32_t aa_plane; } srt_tri_in;

/* Same field order and size (428 B) as `material`, materials/material.cuh:140-148. */
typedef struct {
    float col[3];
    float reflection_fuzz;
    uint32_t material_type;
    float spectral_distribution[SRT_N_CIE_SAMPLES];
    float emission_power;
    float sellmeier_B[3];
    float sellmeier_C[3];
} srt_material;

/* Same field order and size (84 B) as `camera_data`, rendering/rendering.cuh:28-36. */
typedef struct {
    uint32_t width, height;
    float pixel_delta_u[3], pixel_delta_v[3], pixel00_loc[3];
    float defocus_angle;
    float camera_center[3], defocus_disk_u[3], defocus_disk_v[3];
} srt_camera_data;

/* Per-launch counters: a ray = one closest-hit query (bvh::hit call). */
typedef struct {
    uint64_t rays, paths;
    uint64_t node_visits;   /* traversal iterations = paired-child records fetched (V) */
    uint64_t tri_tests;     /* leaf (triangle) tests (T) */
    uint64_t box_tests;
    /* instrumented kernel only: [0] wave-level traversal steps, [1] sum over those steps of lanes that still own
     * work, [2] closest-hit queries answered without traversal because the direction is NaN (the reference walks
     * the whole tree for them and finds nothing), [3] fringe steps, [4] lanes served by them, [5] lanes served by
     * inner steps.  SIMD utilisation of traversal =
     * node_visits / (64 * util[0]).  node_visits / tri_tests / box_tests count the work actually done. */
    uint64_t util[9];   /* [6..8]: wave cycles spent in the shading phase / inner steps / fringe steps */
    uint64_t reserved[2];   /* instrumented: max node visits / max rays of any single pixel */
    /* instrumented kernel only: [0] shading passes (wave level), [1] lanes that shaded a finished query in them,
     * [2] lanes that generated a camera ray, [3] wave-level iterations of the unit-sphere rejection loop */
    uint64_t shade[4];
    /* instrumented kernel only, the tail of a launch: [0] waves, [1] sum of their life times (shader cycles from the start of
     * their state machine to their exit), [2] the longest life, [3] sum of the cycles waves lived on after the pixel queue had
     * run dry for them (drain time: lanes finishing their last pixels).  mean / max life = [1] / ([0] * [2]). */
    uint64_t waves[4];
    /* instrumented kernel only: closest-hit queries that found a triangle (each reads one 48-byte shading record); rays - hits =
     * misses (background look-up) + queries answered without traversal */
    uint64_t hits;
} srt_stats;

typedef struct srt_scene srt_scene;   /* host-side flattened scene (replaces scene_manager's device heap) */
typedef struct srt_ctx srt_ctx;       /* one GPU's renderer (replaces `renderer`, rendering.cuh:39-155) */

/* ---------------------------------------------------------------------------------------------------
 * Host side: scene inputs (no GPU).  Replaces scene_manager::init_world (scene/scene.cu:349-428),
 * create_world_kernel (:22-54) and create_bvh_kernel (:9-20).
 * ------------------------------------------------------------------------------------------------- */
SRT_API const char *srt_version(void);

/* camera::initialize, rendering/camera.cu:7-58 (camera_builder::getCamera, camera_builder.cuh:57-61). */
SRT_API int srt_camera_init(int image_width, int image_height, float vfov, const float lookfrom[3], const float lookat[3],
                            const float vup[3], float defocus_angle, float focus_dist, srt_camera_data *out);

/* Empty scene / one of the built-in scenes (scene/scene.cu:73-226; synthetic ids documented in DESIGN.md).
 * `seed` drives the synthetic scenes' layout PRNG; ignored for the reference scenes. */
SRT_API srt_scene *srt_scene_create(void);
SRT_API srt_scene *srt_scene_builtin(int scene_id, uint64_t seed);
SRT_API void srt_scene_destroy(srt_scene *s);
/* Default camera of a built-in scene (scene/scene.cu:259-320) for the given image size. */
SRT_API int srt_scene_default_camera(const srt_scene *s, int image_width, int image_height, srt_camera_data *out);

/* Scene construction.  srt_scene_set_* replace the whole list. */
SRT_API int srt_scene_set_triangles(srt_scene *s, const srt_tri_in *tris, size_t n);
SRT_API int srt_scene_set_materials(srt_scene *s, const srt_material *mats, size_t m);
SRT_API int srt_scene_set_background(srt_scene *s, const float spectrum[SRT_N_CIE_SAMPLES]);
SRT_API size_t srt_scene_tri_count(const srt_scene *s);
SRT_API size_t srt_scene_material_count(const srt_scene *s);
SRT_API int srt_scene_get_triangles(const srt_scene *s, srt_tri_in *out);       /* raw inputs, original order */
SRT_API int srt_scene_get_materials(const srt_scene *s, srt_material *out);
SRT_API int srt_scene_get_background(const srt_scene *s, float out[SRT_N_CIE_SAMPLES]);
/* tri::init results, 12 floats per triangle: normal(3) D clockwise aa_plane bbox(xmin xmax ymin ymax zmin zmax). */
SRT_API int srt_scene_get_tri_records(const srt_scene *s, float *out);

/* material::compute_spectral_distr (materials/material.cuh:71-84) for table-free colours (grey, white,
 * light, glass); SRT_ERR_UNSUPPORTED for non-grey sRGB (utils/srgb_to_spectrum.cu is absent upstream). */
SRT_API int srt_material_bake(srt_material *m);
/* Quirks of the reference's scene construction (Sellmeier C := B, materials/material.cuh:66-67; Q2: a grey colour's
 * sigmoid coefficient lands in the quadratic slot, color_to_spectrum.cuh:118-120): on (1, default) reproduces the reference as
 * written -- every parity statement refers to that --, 0 builds / bakes what was evidently meant (real Sellmeier C coefficients,
 * grey albedo g -> constant spectrum g).  Process-wide, affects srt_scene_builtin and srt_material_bake / srt_background_spectrum
 * calls made afterwards; returns the previous setting.  The render path itself has no switch. */
SRT_API int srt_set_reference_quirks(int on);
/* dev_srgb_to_spectrum / dev_srgb_to_illuminance_spectrum evaluated from explicit sigmoid coefficients
 * (color/color_to_spectrum.cuh:173-186,204-219): value = [scale * D65n(l)] * sigmoid(c[2] l^2 + c[1] l + c[0]). */
SRT_API int srt_bake_sigmoid_spectrum(const float coeffs[3], float scale, int times_d65, float out[SRT_N_CIE_SAMPLES]);
/* Own Jakob-Hanika style fit of the sigmoid coefficients of a NON-grey sRGB colour (replaces the lookup in the pbrt
 * rgb2spec table of color/color_to_spectrum.cuh:109-151, which the reference mount lacks).  coeffs are in the layout
 * srt_bake_sigmoid_spectrum expects.  Not pinned against the author's table. */
SRT_API int srt_fit_sigmoid_coeffs(const float rgb[3], float coeffs[3]);
/* host srgb_to_illuminance_spectrum for the background (rendering/rendering.cu:324), grey colours only. */
/* The constant tables the path computes with: cmf = 95 rows {x_bar, y_bar, z_bar, normalised D65} (cie_x / cie_y / cie_z /
 * normalized_cie_d65, utils/cie_const.cu:12-122, the host copies the reference uploads into dev_cie_* with
 * cudaMemcpyToSymbol) and the row-major d65_XYZ_to_sRGB matrix (utils/color_const.cu:17-19).  tests/test_ref_tables.py
 * compares them bit for bit with the reference's own arrays compiled from its sources (oracle/Makefile, target `ref`). */
SRT_API int srt_color_tables(float cmf[SRT_N_CIE_SAMPLES * 4], float xyz_to_srgb[9]);
SRT_API int srt_background_spectrum(const float rgb[3], float out[SRT_N_CIE_SAMPLES]);

/* transform::assign_rot_matrix (primitives/transform.cu:4-34): writes the rotation entries of a row-major 3x3 matrix for `axis`
 * 1 = X, 2 = Y, 3 = Z (transform::AXIS, transform.cuh:5-10) into m, which the caller initialised (the reference starts from the
 * identity, tri.cu:97-99); other axis values leave m untouched.  Points are rotated as vec3::matrix_mul does (math/vec3.cuh:80-91):
 * out[i] = m[3i] x + m[3i+1] y + m[3i+2] z.  The matrix the built-in scenes' boxes / pyramid / prism are turned with
 * (scene/scene.cu:115-128,166); tests/test_ref_host.py holds it against the reference's own function compiled from its source. */
SRT_API int srt_rotation_matrix(float theta, int axis, float m[9]);

/* BVH: `mode` SRT_BVH_REFERENCE reproduces create_bvh_kernel (fresh XORWOW(seed), bvh/bvh.cu:206-346);
 * SRT_BVH_SAH is this build's builder.  Either way the node semantics are the reference's: binary tree,
 * one triangle per leaf, leaf box = padded triangle box, internal box = union of children (Q22). */
SRT_API int srt_scene_build_bvh(srt_scene *s, int mode, uint64_t seed);
/* The traversal is the reference's left-first walk, so the child order is part of the tree; SRT_BVH_SAH orders every node's
 * children by distance to the scene's default camera.  For another viewpoint: re-order the built tree for `eye` (the nearer
 * child becomes the left one; topology, boxes and depth unchanged), then srt_upload_scene again. */
SRT_API int srt_scene_order_children(srt_scene *s, const float eye[3]);
/* Topology optimisation of a built tree for throughput-bound launches (host only; no reference counterpart -- the tree is an input
 * of bvh::hit, bvh/bvh.cu:98-166): `passes` rounds of insertion-based optimisation (every subtree is taken out and put back where
 * it adds the least surface area; 3 rounds converge), then boxes, depth and the builder's child order again.  Fewer node records per
 * ray on average; NOT a cheaper worst pixel -- measured slower on launches bound by their longest pixel (few pixels per lane), so it
 * is a call of its own and not part of srt_scene_build_bvh.  Upload the scene afterwards. */
SRT_API int srt_scene_optimise_bvh(srt_scene *s, int passes);
/* 1 when every internal node of the built tree has two leaf children or none ("paired": what SRT_BVH_SAH builds for an even triangle
 * count -- it cuts every span into two even halves -- and what srt_scene_optimise_bvh preserves).  The render launch of such a tree uses
 * the kernel variant whose FRINGE visit carries no box test (a node with one leaf child is the only kind that needs it). */
SRT_API int srt_scene_is_paired(const srt_scene *s);
SRT_API size_t srt_scene_node_count(const srt_scene *s);
SRT_API int srt_scene_bvh_depth(const srt_scene *s);
/* Pre-order dump (same convention as the oracle): left/right = pre-order ranks or -1, prim = original
 * triangle index for leaves else -1, boxes = 6 floats per node (xmin xmax ymin ymax zmin zmax). */
SRT_API int srt_scene_get_bvh(const srt_scene *s, int32_t *left, int32_t *right, int32_t *prim, float *boxes);

/* ---------------------------------------------------------------------------------------------------
 * Device side.  Replaces `renderer` (rendering/rendering.cuh:39-155) + the device half of
 * render_manager::step (rendering/render_manager.cu:3-66).
 * ------------------------------------------------------------------------------------------------- */
/* renderer ctor + hipSetDevice.  Fails with SRT_ERR_NO_DEVICE when no GPU is usable. */
SRT_API int srt_create(int device, srt_ctx **out);
SRT_API void srt_destroy(srt_ctx *ctx);
SRT_API const char *srt_last_error(const srt_ctx *ctx);   /* ctx may be NULL: last global error */

/* Uploads triangles, paired-child BVH records, material spectra and background to HBM
 * (replaces the device-heap world behind bvh** / material*, scene.cuh:163-170).  The BVH must be built. */
SRT_API int srt_upload_scene(srt_ctx *ctx, const srt_scene *s);
/* renderer::assign_cam_data, rendering/rendering.cu:237-242 */
SRT_API int srt_set_camera(srt_ctx *ctx, const srt_camera_data *cam);
/* What the render launch of the uploaded scene looks like (diagnostics / measurement bookkeeping): persistent waves per CU, inner
 * records resident in LDS, whether that is the whole inner tree (kernel variant ALL_CACHED) and whether record references fit 15
 * bits (variant NARROW).  Any pointer may be NULL. */
SRT_API int srt_launch_plan(const srt_ctx *ctx, int *waves_per_cu, int *n_cached, int *all_cached, int *narrow_refs);
/* 1 when the render launch of the uploaded scene uses the PAIRED kernel variant (FRINGE visit without a box test): the tree is paired
 * (srt_scene_is_paired), fits LDS and has 16-bit references. */
SRT_API int srt_launch_paired(const srt_ctx *ctx, int *paired);
/* TEST KNOBS of a context (no reference counterpart; tests/ and tools/ only).  They force a launch through kernel variants the plan
 * would not pick for the scene, so that every instantiated variant is held to the oracle: wide_refs != 0 -> 32-bit child references
 * for small trees too; lds_cache_max >= 0 caps the inner records kept in LDS (0: every inner record from L2; -1: no cap);
 * lane_limit > 0 renders only the first lane_limit pixels of every 8x8 tile (latency experiments; 0: all 64).  Defaults 0 / -1 / 0.
 * Upload the scene again after a change.  The environment never changes them at launch time: SRT_WIDE_REFS, SRT_LDS_CACHE_MAX and
 * SRT_DEBUG_LANE_LIMIT are read ONCE, at srt_create, and ONLY when SRT_TEST_KNOBS=1 is set as well -- a stray variable in a user's
 * shell does not alter the kernel that runs.  srt_get_test_knobs reports what the context uses (from_env: 1 if srt_create took a
 * value from the environment); srt_launch_plan reflects it.  Any out pointer may be NULL. */
SRT_API int srt_set_test_knobs(srt_ctx *ctx, int wide_refs, int lds_cache_max, uint32_t lane_limit);
SRT_API int srt_get_test_knobs(const srt_ctx *ctx, int *wide_refs, int *lds_cache_max, uint32_t *lane_limit, int *from_env);
/* Dynamic LDS bytes of one workgroup of that launch (tables + inner-record cache + traversal stacks): the counterpart of the
 * reference's shared_mem_size (rendering/rendering.cu:290-301), which its run log reports as "shared memory byte size" (:342). */
SRT_API int srt_launch_lds_bytes(const srt_ctx *ctx, size_t *bytes);
/* renderer::init_device_params (rendering/rendering.cu:279-357) + render_manager::init_renderer
 * (render_manager.cu:121-133): threads (tx,ty), grid (bx,by), chunk size, spp, bounce limit, RNG base seed.
 * Allocates the block-linear planar framebuffer and seeds the per-lane RNG states (init_random_states,
 * rendering.cu:120-138: XORWOW(seed + idx)).  spp / bounce_limit are narrowed to 16 bit like the reference (Q17). */
SRT_API int srt_init_device_params(srt_ctx *ctx, uint32_t tx, uint32_t ty, uint32_t bx, uint32_t by, uint32_t chunk_w,
                                   uint32_t chunk_h, uint32_t spp, uint32_t bounce_limit, uint64_t seed);
/* Multi-GPU split: this context renders tiles t with t % world == rank (8x8-pixel tiles of the chunk grid).
 * Seeds depend on the pixel only, so any split gives a bit-identical image.  Default rank 0, world 1. */
SRT_API int srt_set_partition(srt_ctx *ctx, uint32_t rank, uint32_t world);

/* renderer::render(w,h,offx,offy) -> call_render_kernel (rendering.cu:244-277).  Asynchronous on `stream`
 * (a hipStream_t passed as void*, NULL = default stream); the reference's device-wide sync is srt_synchronize.
 * Renders this rank's tiles into the context's compact tile buffer. */
SRT_API int srt_render_chunk(srt_ctx *ctx, uint32_t width, uint32_t height, uint32_t offx, uint32_t offy, void *stream);
SRT_API int srt_synchronize(srt_ctx *ctx);

/* Compact tile buffer of this rank (device memory): three plane GROUPS of tiles_padded * 3 * 64 floats each,
 * [group][tile][plane][lane] -- group 0 = quantised r,g,b (the reference's frame_buffer values, 12 B / pixel), group 1 =
 * unquantised sRGB r,g,b, group 2 = XYZ sums (parity planes).  tiles_padded = ceil(n_tiles/world), so every rank's buffer has
 * the same size.  `planes` = 3 (default): the render kernel writes group 0 only, and group 0 is what a scatter / the multi-GPU
 * gather moves -- what the reference's framebuffer holds; `planes` = 9: the kernel also writes the two parity groups
 * and scatter / gather move all three (parity tests that compare unquantised sRGB or XYZ sums; srt_read_fb_aux needs it).  Set it
 * before srt_render_chunk.  srt_tile_buffer reports n_floats = tiles_padded * planes * 64. */
SRT_API int srt_set_gather_planes(srt_ctx *ctx, uint32_t planes);
SRT_API int srt_tile_buffer(srt_ctx *ctx, void **dev_ptr, size_t *n_floats, uint32_t *tiles_local, uint32_t *tiles_padded);
/* Stream-ordered device-to-device copy of the exchange unit into caller-owned device memory (e.g. the tensor that is
 * handed to the RCCL gather): n_floats as reported by srt_tile_buffer. */
SRT_API int srt_copy_tile_buffer(srt_ctx *ctx, void *dst_dev, void *stream);
/* Scatter gathered exchange units (device pointer, world * tiles_padded * planes * 64 floats, rank-major) into this
 * context's block-linear planar framebuffer (rendering.cu:146-148 layout); with planes == 3 only the quantised planes are
 * written.  With world == 1 pass NULL: the context's own tile buffer (its first `planes` planes) is scattered. */
SRT_API int srt_scatter_tiles(srt_ctx *ctx, const void *dev_gathered, void *stream);

/* renderer::getDevFBr/g/b (rendering.cuh:87-97): device pointers to the block-linear planes (tx*bx*ty*by floats). */
SRT_API int srt_dev_fb(srt_ctx *ctx, void **r, void **g, void **b, size_t *n_floats);
/* The three cudaMemcpyAsync D2H of render_manager::step (render_manager.cu:41-45): block-linear, grid sized. */
SRT_API int srt_read_fb(srt_ctx *ctx, float *r, float *g, float *b);
/* D2H + the un-swizzle of render_manager::update_fb (render_manager.cuh:68-142) done on the device:
 * writes the last rendered chunk into row-major image planes of width image_width at (offx, offy). */
SRT_API int srt_read_fb_rowmajor(srt_ctx *ctx, float *r, float *g, float *b, uint32_t image_width, uint32_t image_height);
/* Parity planes, block-linear: which = 1 unquantised sRGB in [0,1] (value before expand_sRGB), 2 = XYZ sums.  ANY context -- single
 * GPU included -- writes them only when srt_set_gather_planes(ctx, 9) was called before srt_render_chunk: with the default 3 planes
 * the kernel and the scatter move the quantised framebuffer alone and this call returns SRT_ERR_UNSUPPORTED. */
SRT_API int srt_read_fb_aux(srt_ctx *ctx, int which, float *p0, float *p1, float *p2);

/* Scheduling introspection: per-local-tile traversal cost measured by the probe of the last ordered launch (n = tiles_local; n = 2 *
 * tiles_local: followed by the cost of every tile's most expensive pixel -- one pixel is one sequential chain). */
SRT_API int srt_get_tile_costs(srt_ctx *ctx, uint32_t *out, size_t n);
/* Child order of a built tree from a profile of the real rays (no reference counterpart: the reference's bvh::hit, bvh/bvh.cu:98-166,
 * always descends left first, so the order is a property of the tree it is given).  Renders ONE instrumented probe frame of the
 * context's camera (width x height, spp, bounce_limit; srt_set_camera first) in which every closest-hit query notes, at each
 * ancestor of the triangle it found, whether the other child's box lay on the ray beyond the hit -- the rays for which the visiting
 * order decides whether that subtree is pruned -- and swaps the children of every node where the right child won more often (at
 * least min_samples such rays; other nodes keep their order).  The scene is left re-ordered AND uploaded to ctx; topology, boxes,
 * depth unchanged; results can only differ where two triangles tie exactly in t.  Call srt_init_device_params afterwards (the
 * probe used the context's RNG state).  n_swapped may be NULL. */
SRT_API int srt_order_children_by_profile(srt_ctx *ctx, srt_scene *scene, uint32_t width, uint32_t height, uint32_t spp,
                                          uint32_t bounce_limit, uint32_t min_samples, uint32_t *n_swapped);
SRT_API int srt_get_stats(srt_ctx *ctx, srt_stats *out);       /* counters of the last srt_render_chunk */
SRT_API int srt_set_count_traversal(srt_ctx *ctx, int on);     /* 1: instrumented kernel also counts V / T */
/* Instrumented launches only (diagnostics of the tail of a launch): 4 words per persistent wave -- [0] its life time and [1] the
 * moment the pixel queue first came back empty for it (both in units of 256 shader cycles since the wave started; [1] = 2^32-1 if
 * never), [2] its closest-hit queries, [3] the queries of its most expensive pixel.  n_waves <= srt_stats.waves[0]. */
SRT_API int srt_get_wave_debug(srt_ctx *ctx, uint32_t *out, size_t n_waves);
/* Kernel-only time of the last srt_render_chunk in ms, measured with HIP events on its stream. */
SRT_API int srt_last_kernel_ms(srt_ctx *ctx, float *ms);
/* Closest-hit query for explicit rays (bvh::hit, bvh/bvh.cu:98-166) -- KAT entry point.
 * rays: n * 6 floats (origin, direction); out: n * 4 floats (t, tri_index or -1, front_face, mat_index). */
SRT_API int srt_trace_rays(srt_ctx *ctx, const float *rays, size_t n, float *out);
/* Device arithmetic self-test: evaluates op `which` on n operand pairs on the GPU (see DESIGN.md "primitive-op
 * sweep"); used to prove the device's + - * / sqrt fmin cast and srt_powf bits equal the host's. */
SRT_API int srt_device_op_sweep(srt_ctx *ctx, int which, const float *a, const float *b, size_t n, float *out);

SRT_API int srt_ctx_device(const srt_ctx *ctx);                 /* HIP device index of the context */
/* compute units of the context's GPU (a render launch keeps srt_launch_plan's waves_per_cu x 64 pixels in flight on each) */
SRT_API int srt_ctx_cu_count(const srt_ctx *ctx);

/* ---------------------------------------------------------------------------------------------------
 * Multi-GPU.  The reference's caller renders chunk after chunk on ONE GPU
 * (render_manager::step, rendering/render_manager.cu:3-66); a communicator fans one chunk out over W GPUs:
 * rank r renders the 8x8-pixel tiles t with t % W == r, ONE RCCL gather (xGMI inside a node) brings the
 * compact tile buffers to rank 0, which scatters them into its block-linear framebuffer.  The image is
 * bit-identical for every W (per-pixel seeds, rendering.cu:137).  RCCL is loaded (dlopen) on first use.
 * ------------------------------------------------------------------------------------------------- */
#define SRT_COMM_ID_BYTES 128
typedef struct srt_comm srt_comm;
/* One process drives n GPUs: creates one context per device (rank i = devices[i]) and an RCCL communicator
 * over them (ncclCommInitAll), one HIP stream per device. */
SRT_API int srt_comm_init_all(const int *devices, int n, srt_comm **out);
/* One process per GPU (torch.distributed.run, mpirun): rank 0 calls srt_comm_unique_id and hands the 128 bytes
 * to the other ranks by any channel; every rank then wraps its own context.  Sets the context's partition. */
SRT_API int srt_comm_unique_id(unsigned char id[SRT_COMM_ID_BYTES]);
SRT_API int srt_comm_init_rank(srt_ctx *ctx, const unsigned char id[SRT_COMM_ID_BYTES], uint32_t rank, uint32_t world, srt_comm **out);
/* SRT_OK when an RCCL can be loaded in this process, SRT_ERR_UNSUPPORTED (with a message) otherwise.  Cheap and local: a
 * launcher lets every rank call it and agrees on the exchange path BEFORE any rank enters the collective srt_comm_init_rank.
 * An RCCL the host process already mapped (PyTorch's) is re-used; SRT_RCCL_LIB names an explicit library. */
SRT_API int srt_comm_available(void);
/* Planes the gather moves: 3 (default) = the quantised framebuffer, 12 B / pixel; 9 = + the parity planes (72 B / pixel).
 * With one process per GPU EVERY rank must set the same value: every srt_render_frame_multi of such a communicator starts with a
 * 4-byte all-gather of the count each rank's context will really use (whichever call set it, srt_set_gather_planes on the wrapped
 * context included) and fails with SRT_ERR_INVALID on EVERY rank when they differ -- all ranks make the same collective calls, so a
 * disagreement is an error, never a hang in ncclGather.  After a 3-plane frame rank 0's parity planes are not this frame's: srt_read_fb_aux returns
 * SRT_ERR_UNSUPPORTED until a 9-plane frame (or a single-GPU scatter) has written them. */
SRT_API int srt_comm_set_gather_planes(srt_comm *comm, uint32_t planes);
SRT_API void srt_comm_destroy(srt_comm *comm);                 /* destroys the contexts srt_comm_init_all created */
SRT_API const char *srt_comm_last_error(const srt_comm *comm);
SRT_API uint32_t srt_comm_world(const srt_comm *comm);
SRT_API uint32_t srt_comm_local_count(const srt_comm *comm);   /* contexts this process drives */
SRT_API srt_ctx *srt_comm_ctx(srt_comm *comm, uint32_t local_index);
SRT_API srt_ctx *srt_comm_root_ctx(srt_comm *comm);            /* rank 0's context (holds the assembled framebuffer) or NULL */
/* srt_upload_scene / srt_set_camera / srt_init_device_params on every local context (the scene is replicated). */
SRT_API int srt_comm_upload_scene(srt_comm *comm, const srt_scene *s);
SRT_API int srt_comm_set_camera(srt_comm *comm, const srt_camera_data *cam);
SRT_API int srt_comm_init_device_params(srt_comm *comm, uint32_t tx, uint32_t ty, uint32_t bx, uint32_t by, uint32_t chunk_w,
                                        uint32_t chunk_h, uint32_t spp, uint32_t bounce_limit, uint64_t seed);
/* renderer::render(w,h,offx,offy) on W GPUs: render kernels on every local rank's stream, one ncclGather of the tile
 * buffers to rank 0, one scatter kernel there.  Asynchronous; srt_comm_synchronize waits for the local streams.
 * Afterwards rank 0's context answers srt_read_fb / srt_read_fb_rowmajor / srt_dev_fb as after srt_render_chunk. */
SRT_API int srt_render_frame_multi(srt_comm *comm, uint32_t width, uint32_t height, uint32_t offx, uint32_t offy);
SRT_API int srt_comm_synchronize(srt_comm *comm);
/* Closest-hit queries / paths of the last frame summed over the local ranks, and the slowest local render kernel. */
SRT_API int srt_comm_stats(srt_comm *comm, uint64_t *rays, uint64_t *paths, float *max_kernel_ms);
/* Time between the end of a local rank's render kernel and the end of the exchange of the last frame (gather + waiting for
 * the slowest rank + the scatter on rank 0), max over the local ranks, ms; 0 in a 1-rank world. */
SRT_API int srt_comm_last_gather_ms(srt_comm *comm, float *ms);

/* Issue-rate calibration (no reference counterpart: measurement support for bench.py's roofline).  Runs microkernel
 * `kind` (csrc/srt_calib.hip: 0 v_add_f32, 1 v_pk_mul_f32, 2 v_fma_f32, 3 dependent v_add_f32 chain, 4 s_add_u32,
 * 5 v_add_f32 + s_add_u32 interleaved, 6 v_cmp + v_cndmask, 7 / 8 ds_read_b64 linear / random, 9 v_max3_f32,
 * 10 v_add_f32 with 26 of 64 lanes enabled) with one workgroup of waves_per_simd * 256 threads on every CU. */
typedef struct {
    double wave_cycles_mean, wave_cycles_max;   /* s_memtime ticks (shader cycles) a wave spent in its loop */
    double wall_ms;                             /* HIP events around the launch */
    uint64_t instr_per_wave;                    /* instructions of the measured loop body per wave (loop control excluded) */
    uint32_t n_waves, n_cu, waves_per_simd;
    uint32_t reserved;
    double wave_cycles_min;                     /* the SIMD arbitrates by age: with 4 resident waves the oldest finish first, so the
                                                   rate of a SIMD is instructions / wave_cycles_max (or wall x clock), never / mean */
} srt_calibration;
SRT_API int srt_calibrate(srt_ctx *ctx, int kind, uint32_t waves_per_simd, uint32_t iters, srt_calibration *out);

#ifdef __cplusplus
}
#endif
#endif /* SRT_C_API_H */
