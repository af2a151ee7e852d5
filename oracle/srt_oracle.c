/*
 * srt_oracle.c -- CPU ORACLE for the spectral path-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libsrt_hip.so) never links,
 * loads or falls back to anything in oracle/.
 *
 * What it is: a plain-C, scalar, pointer-linked restatement of the reference renderer's
 * per-pixel path (PieSil/CUDA-spectral-ray-tracer, rendering/rendering.cu:151-235 and the
 * device functions it calls).  Every function cites the reference file:line it follows and
 * keeps the reference's IEEE-754 fp32 operation order (no FMA contraction: build with
 * -ffp-contract=off, no fast-math).
 *
 * PARITY PINNING STATUS ("partially pinned"):
 *   - The reference's render path is CUDA-only (every hot function is __device__, needs <curand_kernel.h>
 *     and the CUDA runtime headers).  Those headers do not exist in this image and the build
 *     rules forbid writing stand-ins for them, so that path is UNBUILDABLE here.  The reference ships no tests,
 *     golden vectors or fixtures.
 *   - Two translation units of the reference hold host data and DO compile unmodified with this image's hipcc (-x hip):
 *     utils/cie_const.cu, utils/color_const.cu (constant tables) -- oracle/Makefile's `ref` target builds them from where they lie into
 *     oracle/_ref/libref_tables.so, and tests/test_ref_tables.py requires the CIE / D65 rows and the XYZ->sRGB matrix of
 *     this oracle AND of the product to equal the reference's arrays bit for bit.  The reference's one curand-free DEVICE function,
 *     sellmeier_index (refraction/sellmeier.cu), is compiled the same way into oracle/_ref/libref_sellmeier*.so (with a harness kernel
 *     of ours, oracle/ref_sellmeier_driver.hip) and run on the GPU: orc_sellmeier_index equals it bit for bit when it is built with
 *     -ffp-contract=off; with the compiler's default contraction 2 % of the operands differ (tests/test_ref_tables.py).  Nothing else
 *     of the reference compiles without stand-ins.
 *   - What else pins this oracle: the function-level known answers recorded in SURVEY.md 8(c)/Q1/
 *     Q14/Q23 (outputs of the reference's own functions observed in the survey session),
 *     committed as tests/golden/survey_kats.json and checked by tests/test_oracle_kats.py.
 *   - Whole-image parity against the real CUDA binary is UNPINNED (cannot be produced).
 *   - Third-party arithmetic restated from published definitions, unpinned upstream:
 *       cuRAND XORWOW (curand_init / curand / curand_uniform; CUDA toolkit, version unpinned by
 *       the reference's CMakeLists.txt:2) and CUDA's powf (replaced by srt_powf, see below).
 *       Of XORWOW, the recurrence (orc_rng_next) is held against rocRAND's xorwow_engine::next() -- an independent
 *       implementation that ships with this image's ROCm -- by rocrand_xorwow_pin.hip / tests/test_rocrand_xorwow_pin.py;
 *       curand_init's seed scramble and curand_uniform's float mapping differ from rocRAND's by design and stay restated only.
 *
 * Deliberate, documented deviations from the reference text (all in DESIGN.md):
 *   D1  vec3::random / random_in_unit_disk draw x, then y, then z (reference leaves the order
 *       unspecified: math/vec3.cuh:107-109,242).
 *   D2  tri::aa_plane is an explicit input (sticky across init()) instead of
 *       uninitialised heap memory.
 *   D3  pow() is srt_powf: exp(y*log(x)) evaluated in fp64 by a fixed operation sequence and
 *       rounded once to fp32 (the reference calls CUDA powf, whose bits are unknowable here).
 *   D4  the real material count is used instead of the fixed 32-slot shared copy.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "orc_cie_data.h"

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * constants  (utils/cie_const.cuh:8-12, ray/ray.cuh:12, materials/material.cuh:14-22,
 *             bvh/bvh.cuh:12-13, utils/utility.h:10)
 * ---------------------------------------------------------------------------------------- */
#define N_CIE_SAMPLES 95
#define LAMBDA_MAX 830.0f
#define LAMBDA_MIN 360.0f
#define N_RAY_WAVELENGTHS 7
#define EPSILON 0.0001f
#define MAT_LAMBERTIAN 0u
#define MAT_METALLIC 1u
#define MAT_DIELECTRIC 2u
#define MAT_EMISSIVE 4u
#define MAT_NO_MAT 6u
#define BVH_MAX_DEPTH 64
#define ORC_PI 3.1415926535897932385f

enum { AAP_NONE = 0, AAP_XY = 1, AAP_YZ = 2, AAP_XZ = 3 }; /* primitives/tri.cuh:8-13 */

static float cie_x[N_CIE_SAMPLES], cie_y[N_CIE_SAMPLES], cie_z[N_CIE_SAMPLES];
static float normalized_cie_d65[N_CIE_SAMPLES];
/* utils/color_const.cu:17-19 (d65_XYZ_to_sRGB, row-major) */
static const float d65_XYZ_to_sRGB[9] = { 3.2404542f,  -1.5371385f, -0.4985314f,
                                          -0.9692660f, 1.8760108f,  0.0415560f,
                                          0.0556434f,  -0.2040259f, 1.0572252f };

static void orc_tables_init(void) {
    static int done = 0;
    if (done) return;
    int k = 0;
#define ROW(X, Y, Z, D) cie_x[k] = (float)(X); cie_y[k] = (float)(Y); cie_z[k] = (float)(Z); \
                        normalized_cie_d65[k] = (float)((D) / SRT_D65_NORM_DIVISOR); k++;
    SRT_CIE_ROW_LIST(ROW)
#undef ROW
    done = 1;
}
__attribute__((constructor)) static void orc_ctor(void) { orc_tables_init(); }

/* ------------------------------------------------------------------------------------------
 * types
 * ---------------------------------------------------------------------------------------- */
typedef struct { float e[3]; } vec3;                       /* math/vec3.cuh:11-110 */
typedef struct { float min, max; } interval_t;             /* math/interval.cuh:12-41 */
typedef struct { interval_t x, y, z; } aabb_t;             /* bvh/aabb.cuh:20-114 */

typedef struct {                                           /* primitives/tri.cuh:88-94 */
    vec3 v[3];
    int clockwise;
    int aa_plane;
    uint32_t mat_index;
    aabb_t bbox;
    vec3 normal;
    float D;
} tri_t;

typedef struct bvh_node {                                  /* bvh/bvh.cuh:103-107 */
    struct bvh_node *left, *right;
    int is_leaf;
    tri_t *primitive;
    aabb_t bbox;
} bvh_node;

typedef struct {                                           /* primitives/hit_record.cuh:13-45 */
    vec3 p, normal;
    float t;
    int front_face;
    uint32_t mat_index;
} hit_record;

typedef struct {                                           /* materials/material.cuh:140-148 (428 B) */
    float col[3];
    float reflection_fuzz;
    uint32_t material_type;
    float spectral_distribution[N_CIE_SAMPLES];
    float emission_power;
    float sellmeier_B[3];
    float sellmeier_C[3];
} orc_material;

typedef struct {                                           /* ray/ray.cuh:15-24 */
    vec3 orig, dir;
    uint32_t valid_wavelengths;
    float wavelengths[N_RAY_WAVELENGTHS];
    float power_distr[N_RAY_WAVELENGTHS];
} ray_t;

typedef struct {                                           /* rendering/rendering.cuh:28-36 (84 B) */
    uint32_t width, height;
    float pixel_delta_u[3], pixel_delta_v[3], pixel00_loc[3];
    float defocus_angle;
    float camera_center[3], defocus_disk_u[3], defocus_disk_v[3];
} orc_camera_data;

typedef struct { uint32_t d, v[5]; } orc_rng;              /* cuRAND curandStateXORWOW core words */

typedef struct { float v0[3], v1[3], v2[3]; uint32_t mat_index; uint32_t aa_plane; } orc_tri_in;

typedef struct {
    uint64_t rays, paths, trav_iters, box_tests, tri_tests, max_stack;
} orc_stats;

typedef struct {
    tri_t *tris;         /* storage */
    tri_t **list;        /* permuted by the builder like the reference's tri*[] (bvh/bvh.cu:266) */
    size_t n_tris;
    orc_material *mats;
    size_t n_mats;
    float background[N_CIE_SAMPLES];
    bvh_node *root;
    bvh_node *pool;
    size_t pool_used, pool_cap;
    int valid;
} orc_scene;

/* ------------------------------------------------------------------------------------------
 * RNG: cuRAND XORWOW restated from its published definition (third party).
 * Call sites in the reference: rendering/rendering.cu:137, scene/scene.cu:12-14,
 * utils/cuda_utility.cu:19-49.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_rng_init(uint64_t seed, orc_rng *s) {
    /* curand_init(seed, subsequence = 0, offset = 0): seed scramble only, no skip-ahead. */
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    uint32_t s1 = ((uint32_t)(seed >> 32)) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    s->d = 6615241u + t1 + t0;
    s->v[0] = 123456789u + t0;
    s->v[1] = 362436069u ^ t0;
    s->v[2] = 521288629u + t1;
    s->v[3] = 88675123u ^ t1;
    s->v[4] = 5783321u + t0;
}

ORC_API uint32_t orc_rng_next(orc_rng *s) {
    /* Marsaglia xorwow step + Weyl sequence (curand()). */
    uint32_t t = s->v[0] ^ (s->v[0] >> 2);
    s->v[0] = s->v[1];
    s->v[1] = s->v[2];
    s->v[2] = s->v[3];
    s->v[3] = s->v[4];
    s->v[4] = (s->v[4] ^ (s->v[4] << 4)) ^ (t ^ (t << 1));
    s->d += 362437u;
    return s->v[4] + s->d;
}

/* utils/cuda_utility.cu:19-26 -> curand_uniform: (0,1] */
ORC_API float orc_random_float(orc_rng *s) {
    uint32_t x = orc_rng_next(s);
    return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

/* utils/cuda_utility.cu:28-41 */
ORC_API float orc_random_float_range(float min, float max, orc_rng *s) {
    float range_width = max - min;
    float random = orc_random_float(s);
    float result = random * range_width + min;
    return result;
}

/* utils/cuda_utility.cu:44-49 */
ORC_API int orc_random_int(int min, int max, orc_rng *s) {
    float random_float = orc_random_float_range((float)(min - 1), (float)(max - 1), s);
    return (int)ceilf(random_float);
}

/* ------------------------------------------------------------------------------------------
 * srt_powf (deviation D3): the project-wide definition of pow() on the path.
 * Call sites in the reference: materials/material.cu:48, color/color.cu:19.
 * Spec (all fp64, no contraction):  y == 5: ((x*x)*(x*x))*x.  Otherwise x = m*2^e with m in (sqrt(.5), sqrt(2)];
 *   f=(m-1)/(m+1); s=f*f; log(m) = 2 f (1 + s/3 + ... + s^12/25)  (Horner);
 *   A = y*e (exact); z = y*log m; k = floor((A + z*LOG2E) + .5); d = A - k (exact);
 *   r = (d*LN2_HI + z) + d*LN2_LO;
 *   exp(r) = sum_{n<=14} r^n/n! (Horner); result = (float)(exp(r) * 2^k).
 * ---------------------------------------------------------------------------------------- */
ORC_API float orc_powf(float xf, float yf) {
    if (xf != xf || yf != yf) return xf + yf;
    if (yf == 0.0f || xf == 1.0f) return 1.0f;
    if (xf == 0.0f) return yf > 0.0f ? 0.0f : INFINITY;
    if (xf < 0.0f) return NAN;
    if (isinf(xf)) return yf > 0.0f ? INFINITY : 0.0f;
    double x = (double)xf, y = (double)yf;
    if (yf == 5.0f) {
        /* Schlick's (1-cos)^5 (materials/material.cu:48), the only integer exponent on the per-ray path: x^2 is exact in
         * fp64 (48 bits), the two further products round once each, so the result is within 2.2e-16 of x^5 before the
         * single rounding to fp32 -- the same accuracy class as the general branch at a fraction of the cost. */
        const double x2 = x * x;
        return (float)((x2 * x2) * x);
    }
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ffu) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    memcpy(&m, &bits, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    double f = (m - 1.0) / (m + 1.0);
    double s = f * f;
    double p = 1.0 / 25.0;
    p = p * s + 1.0 / 23.0;
    p = p * s + 1.0 / 21.0;
    p = p * s + 1.0 / 19.0;
    p = p * s + 1.0 / 17.0;
    p = p * s + 1.0 / 15.0;
    p = p * s + 1.0 / 13.0;
    p = p * s + 1.0 / 11.0;
    p = p * s + 1.0 / 9.0;
    p = p * s + 1.0 / 7.0;
    p = p * s + 1.0 / 5.0;
    p = p * s + 1.0 / 3.0;
    p = p * s + 1.0;
    double lg = 2.0 * f * p;
    double A = y * (double)e;                 /* exact: 24-bit * 11-bit */
    double z = y * lg;
    double kd = floor((A + z * 1.4426950408889634) + 0.5);
    if (kd > 1000.0) return INFINITY;
    if (kd < -1000.0) return 0.0f;
    double dd = A - kd;                        /* exact */
    double r = (dd * 0.693147180369123816490 + z) + dd * 1.90821492927058770002e-10;
    double q = 1.0 / 87178291200.0;
    q = q * r + 1.0 / 6227020800.0;
    q = q * r + 1.0 / 479001600.0;
    q = q * r + 1.0 / 39916800.0;
    q = q * r + 1.0 / 3628800.0;
    q = q * r + 1.0 / 362880.0;
    q = q * r + 1.0 / 40320.0;
    q = q * r + 1.0 / 5040.0;
    q = q * r + 1.0 / 720.0;
    q = q * r + 1.0 / 120.0;
    q = q * r + 1.0 / 24.0;
    q = q * r + 1.0 / 6.0;
    q = q * r + 0.5;
    q = q * r + 1.0;
    q = q * r + 1.0;
    uint64_t kb = (uint64_t)((int64_t)kd + 1023) << 52;
    double two_k;
    memcpy(&two_k, &kb, 8);
    return (float)(q * two_k);
}

/* ------------------------------------------------------------------------------------------
 * vec3 algebra  (math/vec3.cuh)
 * ---------------------------------------------------------------------------------------- */
static inline vec3 V(float a, float b, float c) { vec3 r = { { a, b, c } }; return r; }
static inline vec3 vadd(vec3 u, vec3 v) { return V(u.e[0] + v.e[0], u.e[1] + v.e[1], u.e[2] + v.e[2]); } /* :119 */
static inline vec3 vsub(vec3 u, vec3 v) { return V(u.e[0] - v.e[0], u.e[1] - v.e[1], u.e[2] - v.e[2]); } /* :124 */
static inline vec3 vscale(float t, vec3 v) { return V(t * v.e[0], t * v.e[1], t * v.e[2]); }             /* :134 */
static inline vec3 vneg(vec3 v) { return V(-v.e[0], -v.e[1], -v.e[2]); }                                 /* :36 */
static inline vec3 vdiv(vec3 v, float t) { return vscale(1 / t, v); }                                    /* :144-147 */
static inline float vdot(vec3 u, vec3 v) { return u.e[0] * v.e[0] + u.e[1] * v.e[1] + u.e[2] * v.e[2]; } /* :149 */
static inline vec3 vcross(vec3 u, vec3 v) {                                                              /* :155 */
    return V(u.e[1] * v.e[2] - u.e[2] * v.e[1], u.e[2] * v.e[0] - u.e[0] * v.e[2], u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}
static inline float vlen2(vec3 v) { float x = v.e[0], y = v.e[1], z = v.e[2]; return x * x + y * y + z * z; } /* :66-72 */
static inline float vlen(vec3 v) { return sqrtf(vlen2(v)); }                                                /* :75-78 */
static inline vec3 vunit(vec3 v) { return vdiv(v, vlen(v)); }                                               /* :161 */
static inline int vnear_zero(vec3 v) {                                                                      /* :93-98 */
    float s = 1e-8f;
    return (fabsf(v.e[0]) < s) && (fabsf(v.e[1]) < s) && (fabsf(v.e[2]) < s);
}
static inline vec3 vmatmul(vec3 v, const float *m) {                                                        /* :80-91 */
    float e0 = v.e[0], e1 = v.e[1], e2 = v.e[2];
    return V((m[0] * e0) + (m[1] * e1) + (m[2] * e2), (m[3] * e0) + (m[4] * e1) + (m[5] * e2),
             (m[6] * e0) + (m[7] * e1) + (m[8] * e2));
}
static inline vec3 vreflect(vec3 v, vec3 n) { return vsub(v, vscale(2 * vdot(v, n), n)); }                  /* :180-183 */
static inline vec3 vrefract(vec3 uv, vec3 n, float etai_over_etat) {                                       /* :199-205 */
    float cos_theta = fminf(vdot(vneg(uv), n), 1.0f);
    vec3 r_out_perp = vscale(etai_over_etat, vadd(uv, vscale(cos_theta, n)));
    vec3 r_out_parallel = vscale(-sqrtf(fabsf(1.0f - vlen2(r_out_perp))), n);
    return vadd(r_out_perp, r_out_parallel);
}
/* math/vec3.cuh:106-109 with deviation D1 (x, y, z drawn in that order) */
static inline vec3 vrandom_range(float min, float max, orc_rng *s) {
    float x = orc_random_float_range(min, max, s);
    float y = orc_random_float_range(min, max, s);
    float z = orc_random_float_range(min, max, s);
    return V(x, y, z);
}
static vec3 random_in_unit_sphere(orc_rng *s) {                                                            /* :210-218 */
    for (;;) {
        vec3 p = vrandom_range(-1, 1, s);
        if (vlen2(p) < 1.0f) return p;
    }
}
static vec3 random_unit_vector(orc_rng *s) { return vunit(random_in_unit_sphere(s)); }                     /* :221-227 */
static vec3 random_in_unit_disk(orc_rng *s) {                                                              /* :240-246, D1 */
    for (;;) {
        float x = orc_random_float_range(-1, 1, s);
        float y = orc_random_float_range(-1, 1, s);
        vec3 p = V(x, y, 0);
        if (vlen2(p) < 1.0f) return p;
    }
}

/* ------------------------------------------------------------------------------------------
 * spectrum / colour  (spectrum/spectrum.cu, color/color.cu)
 * ---------------------------------------------------------------------------------------- */
/* spectrum/spectrum.cu:11-22 */
ORC_API float orc_spectrum_interp(const float *spectrum, float lambda, int n_samples) {
    lambda -= LAMBDA_MIN;
    lambda *= ((float)n_samples - 1) / (LAMBDA_MAX - LAMBDA_MIN);
    int offset = (int)lambda;
    if (offset < 0) offset = 0;
    if (offset > n_samples - 2) offset = n_samples - 2;
    float weight = lambda - (float)offset;
    return (1.0f - weight) * spectrum[offset] + weight * spectrum[offset + 1];
}

/* spectrum/spectrum.cu:31-48 */
static void init_hero_wavelength(float *spectrum, uint32_t n_lambdas, orc_rng *s) {
    float step = (LAMBDA_MAX - LAMBDA_MIN) / (float)n_lambdas;
    float hero = orc_random_float_range(LAMBDA_MIN, LAMBDA_MAX, s);
    spectrum[0] = hero;
    float lambda = hero;
    for (uint32_t i = 1; i < n_lambdas; i++) {
        lambda += step;
        if (lambda > LAMBDA_MAX) {
            float remainder = lambda - LAMBDA_MAX;
            lambda = LAMBDA_MIN + remainder;
        }
        spectrum[i] = lambda;
    }
}

/* ray/ray.cuh:37-44,47-51 */
static void ray_init(ray_t *r, vec3 origin, vec3 direction, orc_rng *s) {
    r->orig = origin;
    r->dir = direction;
    init_hero_wavelength(r->wavelengths, N_RAY_WAVELENGTHS, s);
    for (int i = 0; i < N_RAY_WAVELENGTHS; i++) r->power_distr[i] = 1.0f;
    r->valid_wavelengths = N_RAY_WAVELENGTHS;
}
/* ray/ray.cuh:31-34 */
static inline vec3 ray_at(const ray_t *r, float t) { return vadd(r->orig, vscale(t, r->dir)); }
/* ray/ray.cuh:59-69 */
static void ray_mul_spectrum(ray_t *r, const float *spectral_distr, uint32_t n_samples) {
    for (uint32_t i = 0; i < r->valid_wavelengths; i++) {
        float lambda = r->wavelengths[i];
        float weight = orc_spectrum_interp(spectral_distr, lambda, (int)n_samples);
        r->power_distr[i] *= weight;
    }
}

/* color/color.cu:88-104 */
static vec3 dev_spectrum_to_XYZ(const float *spectrum, const float *power_distribution, uint32_t n_total_samples,
                                uint32_t n_nonzero_samples) {
    float delta_lambda = (LAMBDA_MAX - LAMBDA_MIN) / (float)n_total_samples;
    float x = 0.0f, y = 0.0f, z = 0.0f;
    for (uint32_t i = 0; i < n_nonzero_samples; i++) {
        float lambda = spectrum[i];
        float power = power_distribution[i];
        x += orc_spectrum_interp(cie_x, lambda, N_CIE_SAMPLES) * power * delta_lambda;
        y += orc_spectrum_interp(cie_y, lambda, N_CIE_SAMPLES) * power * delta_lambda;
        z += orc_spectrum_interp(cie_z, lambda, N_CIE_SAMPLES) * power * delta_lambda;
    }
    return V(x, y, z);
}

/* color/color.cu:15-22 (pow -> srt_powf, D3) */
ORC_API float orc_correct_channel(float value) {
    float res = value < 0.0f ? 0.0f
              : (value < 0.0031308f ? 12.92f * value
              : (value < 1.0f ? ((1.055f * orc_powf(value, 0.416666f)) - 0.055f) : 1.0f));
    return res;
}
/* color/color.cu:35-41 */
static vec3 XYZ_to_sRGB(vec3 xyz, const float *m) {
    vec3 nc = vmatmul(xyz, m);
    return V(orc_correct_channel(nc.e[0]), orc_correct_channel(nc.e[1]), orc_correct_channel(nc.e[2]));
}
/* color/color.cu:43-49 */
static vec3 expand_sRGB(vec3 c) {
    return V((float)(int)(c.e[0] * 255.99f), (float)(int)(c.e[1] * 255.99f), (float)(int)(c.e[2] * 255.99f));
}

/* ------------------------------------------------------------------------------------------
 * refraction / materials  (refraction/sellmeier.cu, materials/material.cu)
 * ---------------------------------------------------------------------------------------- */
/* refraction/sellmeier.cu:11-22 */
ORC_API float orc_sellmeier_index(const float b[3], const float c[3], float lambda) {
    lambda *= 1e-3f;
    float lambda_squared = lambda * lambda;
    float index = 1.0f + (b[0] * lambda_squared) / (lambda_squared - c[0]) +
                  (b[1] * lambda_squared) / (lambda_squared - c[1]) +
                  (b[2] * lambda_squared) / (lambda_squared - c[2]);
    index = sqrtf(index);
    return index;
}

/* materials/material.cu:39-53 */
ORC_API float orc_reflectance(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * orc_powf(1.0f - cosine, 5.0f);
}

/* materials/material.cu:8-19 */
static int lambertian_scatter(const hit_record *rec, vec3 *scatter_direction, orc_rng *s) {
    *scatter_direction = vadd(rec->normal, random_unit_vector(s));
    if (vnear_zero(*scatter_direction)) *scatter_direction = rec->normal;
    return 1;
}
/* materials/material.cu:22-37 */
static int reflection_scatter(float mat_fuzz, vec3 unit_in_direction, const hit_record *rec, vec3 *scattered_direction,
                              orc_rng *s) {
    vec3 reflected = vreflect(unit_in_direction, rec->normal);
    *scattered_direction = vadd(reflected, vscale(mat_fuzz, random_unit_vector(s)));
    return vdot(*scattered_direction, rec->normal) > 0;
}
/* materials/material.cu:102-136 */
static int refraction_scatter(float mat_ir, const hit_record *rec, float *epsilon_correction_sign, vec3 *scatter_direction,
                              vec3 unit_in_direction, orc_rng *s) {
    float refraction_ratio = rec->front_face ? (1.0f / mat_ir) : mat_ir;
    float cos_theta = fminf(vdot(vneg(unit_in_direction), rec->normal), 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    /* short-circuit: the uniform is only drawn when the first operand is false (material.cu:114) */
    int cannot_refract = refraction_ratio * sin_theta > 1.0f || orc_reflectance(cos_theta, refraction_ratio) > orc_random_float(s);
    if (cannot_refract) {
        *scatter_direction = vreflect(unit_in_direction, rec->normal);
    } else {
        *scatter_direction = vrefract(unit_in_direction, rec->normal, refraction_ratio);
        *epsilon_correction_sign = -1.0f;
    }
    return !cannot_refract;
}

/* materials/material.cu:55-100 */
static int material_scatter(const orc_material *mat, ray_t *r_in, const hit_record *rec, orc_rng *s) {
    vec3 scatter_direction = V(0.f, 0.f, 0.f);
    float epsilon_correction_sign = 1.0f;
    int did_scatter = 1;
    vec3 unit_in_direction = vunit(r_in->dir);

    switch (mat->material_type) {
    case MAT_METALLIC:
        did_scatter = reflection_scatter(mat->reflection_fuzz, unit_in_direction, rec, &scatter_direction, s);
        if (!did_scatter) r_in->valid_wavelengths = 0;
        break;
    case MAT_DIELECTRIC: {
        float ir = orc_sellmeier_index(mat->sellmeier_B, mat->sellmeier_C, r_in->wavelengths[0]);
        int refracted = refraction_scatter(ir, rec, &epsilon_correction_sign, &scatter_direction, unit_in_direction, s);
        if (refracted) r_in->valid_wavelengths = 1;
        break;
    }
    case MAT_EMISSIVE:
        did_scatter = 0;
        break;
    case MAT_LAMBERTIAN:
    default:
        lambertian_scatter(rec, &scatter_direction, s);
        break;
    }

    ray_mul_spectrum(r_in, mat->spectral_distribution, N_CIE_SAMPLES);
    r_in->orig = vadd(rec->p, vscale(epsilon_correction_sign * EPSILON, rec->normal));
    r_in->dir = scatter_direction;
    return did_scatter;
}

/* ------------------------------------------------------------------------------------------
 * aabb / tri  (bvh/aabb.cu, bvh/aabb.cuh, primitives/tri.cu)
 * ---------------------------------------------------------------------------------------- */
static inline const interval_t *aabb_axis(const aabb_t *b, int n) { return n == 1 ? &b->y : (n == 2 ? &b->z : &b->x); } /* aabb.cuh:78-90 */

/* bvh/aabb.cu:7-40 */
static int aabb_hit(const aabb_t *box, const ray_t *r, float min, float max) {
    for (int a = 0; a < 3; a++) {
        float inverseDir = 1 / r->dir.e[a];
        float orig = r->orig.e[a];
        float t0, t1;
        if (inverseDir >= 0) {
            t0 = (aabb_axis(box, a)->min - orig) * inverseDir;
            t1 = (aabb_axis(box, a)->max - orig) * inverseDir;
        } else {
            t1 = (aabb_axis(box, a)->min - orig) * inverseDir;
            t0 = (aabb_axis(box, a)->max - orig) * inverseDir;
        }
        if (t0 > min) min = t0;
        if (t1 < max) max = t1;
        if (max <= min) return 0;
    }
    return 1;
}

static interval_t itv(float mn, float mx) { interval_t i = { mn, mx }; return i; }
/* bvh/aabb.cuh:30-34, math/interval.cuh:19-20 */
static aabb_t aabb_union(const aabb_t *a, const aabb_t *b) {
    aabb_t r;
    r.x = itv(fminf(a->x.min, b->x.min), fmaxf(a->x.max, b->x.max));
    r.y = itv(fminf(a->y.min, b->y.min), fmaxf(a->y.max, b->y.max));
    r.z = itv(fminf(a->z.min, b->z.min), fmaxf(a->z.max, b->z.max));
    return r;
}
/* bvh/aabb.cuh:48-57 */
static aabb_t aabb_of_tri(vec3 v1, vec3 v2, vec3 v3) {
    aabb_t r;
    r.x = itv(fminf(v1.e[0], fminf(v2.e[0], v3.e[0])), fmaxf(v1.e[0], fmaxf(v2.e[0], v3.e[0])));
    r.y = itv(fminf(v1.e[1], fminf(v2.e[1], v3.e[1])), fmaxf(v1.e[1], fmaxf(v2.e[1], v3.e[1])));
    r.z = itv(fminf(v1.e[2], fminf(v2.e[2], v3.e[2])), fmaxf(v1.e[2], fmaxf(v2.e[2], v3.e[2])));
    return r;
}
/* bvh/aabb.cuh:92-102, math/interval.cuh:54-63 */
static interval_t pad_axis(interval_t i, float delta) {
    if ((i.max - i.min) >= delta) return i;
    float padding = delta / 2;
    return itv(i.min - padding, i.max + padding);
}
static aabb_t aabb_pad(aabb_t b) {
    float delta = 0.0001f;
    aabb_t r;
    r.x = pad_axis(b.x, delta);
    r.y = pad_axis(b.y, delta);
    r.z = pad_axis(b.z, delta);
    return r;
}

/* primitives/tri.cu:153-182 */
static float double_signed_area_2D(const tri_t *t, vec3 v1, vec3 v2, vec3 v3) {
    uint32_t w_axis, h_axis;
    switch (t->aa_plane) {
    case AAP_YZ: w_axis = 1; h_axis = 2; break;
    case AAP_XZ: w_axis = 0; h_axis = 2; break;
    case AAP_XY:
    default:     w_axis = 0; h_axis = 1;
    }
    return (v1.e[w_axis] - v3.e[w_axis]) * (v2.e[h_axis] - v3.e[h_axis]) -
           (v2.e[w_axis] - v3.e[w_axis]) * (v1.e[h_axis] - v3.e[h_axis]);
}
/* primitives/tri.cu:121-128 */
static int is_interior_faster(const tri_t *t, vec3 p) {
    float a1 = double_signed_area_2D(t, p, t->v[0], t->v[1]);
    float a2 = double_signed_area_2D(t, p, t->v[1], t->v[2]);
    float a3 = double_signed_area_2D(t, p, t->v[2], t->v[0]);
    return t->clockwise ? (a1 >= 0.f && a2 >= 0.f && a3 >= 0.f) : (a1 <= 0.f && a2 <= 0.f && a3 <= 0.f);
}
/* primitives/tri.cu:47-84 (aa_plane keeps its previous value unless the normal is axis aligned: Q12/D2) */
static void tri_init(tri_t *t) {
    vec3 n = vcross(vsub(t->v[1], t->v[0]), vsub(t->v[2], t->v[0]));
    t->normal = vunit(n);
    int perp_x = fabsf(vdot(t->normal, V(1.f, 0.f, 0.f))) < 1e-8f;
    int perp_y = fabsf(vdot(t->normal, V(0.f, 1.f, 0.f))) < 1e-8f;
    int perp_z = fabsf(vdot(t->normal, V(0.f, 0.f, 1.f))) < 1e-8f;
    if (perp_y && perp_z) t->aa_plane = AAP_YZ;
    else if (perp_x && perp_z) t->aa_plane = AAP_XZ;
    else if (perp_x && perp_y) t->aa_plane = AAP_XY;
    t->D = vdot(t->normal, t->v[0]);
    t->clockwise = double_signed_area_2D(t, t->v[0], t->v[1], t->v[2]) >= 0; /* tri.cuh:107-110 */
    t->bbox = aabb_pad(aabb_of_tri(t->v[0], t->v[1], t->v[2]));              /* tri.cuh:54-57 */
}

/* primitives/tri.cu:3-45 + hit_record.cuh:31-44 */
static int tri_hit(const tri_t *t, const ray_t *r, float min, float max, hit_record *rec) {
    float denom = vdot(t->normal, r->dir);
    if (fabsf(denom) < 1e-8f) return 0;
    float tt = (t->D - vdot(t->normal, r->orig)) / denom;
    if (!(min <= tt && tt <= max)) return 0;                 /* interval::contains, interval.cuh:44-46 */
    vec3 intersection = ray_at(r, tt);
    if (!is_interior_faster(t, intersection)) return 0;
    rec->t = tt;
    rec->p = intersection;
    rec->mat_index = t->mat_index;
    rec->front_face = vdot(r->dir, t->normal) < 0;
    rec->normal = rec->front_face ? t->normal : vneg(t->normal);
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * BVH: builder (bvh/bvh.cu:7-71,206-346) and traversal (bvh/bvh.cu:73-166)
 * ---------------------------------------------------------------------------------------- */
static bvh_node *node_new(orc_scene *sc, int is_leaf) {        /* bvh.cuh:29-33 */
    bvh_node *n = &sc->pool[sc->pool_used++];
    n->left = n->right = NULL;
    n->is_leaf = is_leaf;
    n->primitive = NULL;
    n->bbox.x = n->bbox.y = n->bbox.z = itv(+FLT_MAX, -FLT_MAX); /* empty interval, interval.cuh:14 */
    return n;
}
static inline aabb_t node_bounding_box(const bvh_node *n) { return n->is_leaf ? n->primitive->bbox : n->bbox; } /* bvh.cuh:56-59 */

/* bvh.cuh:180-184: comparator = bbox MIN on the axis, strict < */
static int box_compare(const tri_t *a, const tri_t *b, int axis_index) {
    return aabb_axis(&a->bbox, axis_index)->min < aabb_axis(&b->bbox, axis_index)->min;
}
static void swap_tri(tri_t **o, int a, int b) { tri_t *t = o[a]; o[a] = o[b]; o[b] = t; }   /* bvh.cu:7-12 */
static int partition(tri_t **o, int l, int h, int axis) {                                   /* bvh.cu:14-31 */
    if (l == h) return l;
    tri_t *x = o[h];
    int i = l - 1;
    for (int j = l; j < h; j++) {
        if (box_compare(o[j], x, axis)) { i++; swap_tri(o, i, j); }
    }
    swap_tri(o, i + 1, h);
    return i + 1;
}
static void quicksort_primitives(tri_t **o, int start, int end, int axis) {                 /* bvh.cu:33-71 */
    int *stack = (int *)malloc(sizeof(int) * (size_t)(end - start + 1 + 2));
    int top = -1;
    stack[++top] = start;
    stack[++top] = end;
    while (top >= 0) {
        end = stack[top--];
        start = stack[top--];
        int p = partition(o, start, end, axis);
        if (p - 1 > start) { stack[++top] = start; stack[++top] = p - 1; }
        if (p + 1 < end)   { stack[++top] = p + 1; stack[++top] = end; }
    }
    free(stack);
}

static int build_bvh(orc_scene *sc, orc_rng *rs) {                                          /* bvh.cu:206-309 */
    size_t st_start[BVH_MAX_DEPTH], st_end[BVH_MAX_DEPTH];
    bvh_node *node_stack[BVH_MAX_DEPTH];
    int tos = -1;
    sc->root = node_new(sc, 0);
    tos++;
    st_start[tos] = 0; st_end[tos] = sc->n_tris; node_stack[tos] = sc->root;
    tri_t **src = sc->list;
    while (tos >= 0) {
        size_t cs = st_start[tos], ce = st_end[tos];
        bvh_node *node = node_stack[tos];
        tos--;
        size_t span = ce - cs;
        if (span > 0) {
            if (span == 1) {
                node->is_leaf = 1;
                node->left = node->right = NULL;
                node->primitive = src[cs];
            } else {
                int axis = orc_random_int(0, 2, rs);          /* only ever 0 or 1: SURVEY Q14 */
                if (span == 2) {
                    node->left = node_new(sc, 1);
                    node->right = node_new(sc, 1);
                    if (box_compare(src[cs], src[cs + 1], axis)) {
                        node->left->primitive = src[cs];
                        node->right->primitive = src[cs + 1];
                    } else {
                        node->left->primitive = src[cs + 1];
                        node->right->primitive = src[cs];
                    }
                } else {
                    quicksort_primitives(src, (int)cs, (int)(ce - 1), axis);
                    node->left = node_new(sc, 0);
                    node->right = node_new(sc, 0);
                    size_t mid = cs + span / 2;
                    tos++;
                    if (tos >= BVH_MAX_DEPTH) return 0;
                    st_start[tos] = cs; st_end[tos] = mid; node_stack[tos] = node->left;
                    tos++;
                    if (tos >= BVH_MAX_DEPTH) return 0;
                    st_start[tos] = mid; st_end[tos] = ce; node_stack[tos] = node->right;
                }
            }
        }
    }
    return 1;
}

/* bvh.cu:311-346 + bvh.cuh:61-73: every internal box = union of its children's boxes (post-order).
 * The union is exact (fmin/fmax), so recursion order cannot change values. */
static void build_nodes_bboxes(bvh_node *n) {
    if (n->is_leaf) return;
    /* iterative post-order to survive degenerate (very deep) user-supplied trees */
    size_t cap = 64, top = 0;
    bvh_node **st = (bvh_node **)malloc(cap * sizeof(*st));
    char *vis = (char *)malloc(cap);
    st[top] = n; vis[top] = 0; top++;
    while (top) {
        bvh_node *c = st[top - 1];
        if (c->is_leaf) { top--; continue; }
        if (!vis[top - 1]) {
            vis[top - 1] = 1;
            if (top + 2 > cap) { cap *= 2; st = (bvh_node **)realloc(st, cap * sizeof(*st)); vis = (char *)realloc(vis, cap); }
            if (c->right) { st[top] = c->right; vis[top] = 0; top++; }
            if (c->left) { st[top] = c->left; vis[top] = 0; top++; }
        } else {
            if (c->left && c->right) {
                aabb_t a = node_bounding_box(c->left), b = node_bounding_box(c->right);
                c->bbox = aabb_union(&a, &b);
            } else if (c->left) c->bbox = node_bounding_box(c->left);
            else if (c->right) c->bbox = node_bounding_box(c->right);
            top--;
        }
    }
    free(st); free(vis);
}

/* bvh.cu:73-76 */
static inline int node_hit(const bvh_node *n, const ray_t *r, float min, float max, hit_record *rec, orc_stats *st) {
    if (n->is_leaf) { st->tri_tests++; return tri_hit(n->primitive, r, min, max, rec); }
    st->box_tests++;
    return aabb_hit(&n->bbox, r, min, max);
}

/* bvh.cu:98-166 */
static int bvh_hit(const ray_t *r, float min, float max, hit_record *rec, const bvh_node *root, orc_stats *st) {
    int hit_anything = 0;
    float closest_so_far = max;
    const bvh_node *stack[256];
    const bvh_node **stack_ptr = stack;
    *stack_ptr++ = NULL;
    const bvh_node *node = root;
    st->rays++;
    if (node->is_leaf) {
        if (node_hit(node, r, min, closest_so_far, rec, st)) {
            hit_anything = 1;
            closest_so_far = rec->t;
        }
    } else do {
        bvh_node *child_l = node->left;
        bvh_node *child_r = node->right;
        st->trav_iters++;
        hit_record temp_rec;
        int hits_l = child_l != NULL && node_hit(child_l, r, min, closest_so_far, &temp_rec, st);
        if (hits_l && child_l->is_leaf) {
            hit_anything = 1;
            closest_so_far = temp_rec.t;
            *rec = temp_rec;
        }
        int hits_r = child_r != NULL && node_hit(child_r, r, min, closest_so_far, &temp_rec, st);
        if (hits_r && child_r->is_leaf) {
            hit_anything = 1;
            closest_so_far = temp_rec.t;
            *rec = temp_rec;
        }
        int traverse_l = child_l != NULL && (hits_l && !child_l->is_leaf);
        int traverse_r = child_r != NULL && (hits_r && !child_r->is_leaf);
        if (!traverse_l && !traverse_r)
            node = *--stack_ptr;
        else {
            node = traverse_l ? child_l : child_r;
            if (traverse_l && traverse_r) {
                *stack_ptr++ = child_r;
                uint64_t depth = (uint64_t)(stack_ptr - stack) - 1;
                if (depth > st->max_stack) st->max_stack = depth;
            }
        }
    } while (node != NULL);
    return hit_anything;
}

/* ------------------------------------------------------------------------------------------
 * per-pixel driver  (rendering/rendering.cu:12-87,140-235)
 * ---------------------------------------------------------------------------------------- */
static vec3 v3(const float *p) { return V(p[0], p[1], p[2]); }

/* rendering.cu:49-56 */
static vec3 pixel_sample_square(vec3 du, vec3 dv, orc_rng *s) {
    float px = -0.5f + orc_random_float(s);
    float py = -0.5f + orc_random_float(s);
    return vadd(vscale(px, du), vscale(py, dv));
}
/* rendering.cu:42-47 */
static vec3 defocus_disk_sample(vec3 center, vec3 disk_u, vec3 disk_v, orc_rng *s) {
    vec3 p = random_in_unit_disk(s);
    return vadd(vadd(center, vscale(p.e[0], disk_u)), vscale(p.e[1], disk_v));
}
/* rendering.cu:66-87 */
static void get_ray(ray_t *out, uint32_t i, uint32_t j, const orc_camera_data *c, orc_rng *s) {
    vec3 du = v3(c->pixel_delta_u), dv = v3(c->pixel_delta_v);
    vec3 pixel_center = vadd(vadd(v3(c->pixel00_loc), vscale((float)i, du)), vscale((float)j, dv));
    vec3 pixel_sample = vadd(pixel_center, pixel_sample_square(du, dv, s));
    vec3 ray_origin = (c->defocus_angle <= 0.0f)
                          ? v3(c->camera_center)
                          : defocus_disk_sample(v3(c->camera_center), v3(c->defocus_disk_u), v3(c->defocus_disk_v), s);
    vec3 ray_direction = vsub(pixel_sample, ray_origin);
    ray_init(out, ray_origin, ray_direction, s);
}

/* rendering.cu:12-40 */
static void ray_bounce(const orc_scene *sc, ray_t *r, uint32_t bounce_limit, orc_rng *s, orc_stats *st) {
    hit_record rec;
    for (uint32_t n = 0; n < bounce_limit; n++) {
        if (!bvh_hit(r, 0.0f, FLT_MAX, &rec, sc->root, st)) {
            ray_mul_spectrum(r, sc->background, N_CIE_SAMPLES);
            return;
        }
        const orc_material *mat = &sc->mats[rec.mat_index];      /* D4 */
        if (!material_scatter(mat, r, &rec, s)) return;
    }
    r->valid_wavelengths = 0;
}

typedef struct {
    const orc_scene *sc;
    const orc_camera_data *cam;
    uint32_t spp, bounce_limit;
    uint32_t tx, ty, bx, by;         /* block / grid dims (render_manager.cu:93-97) */
    uint32_t width, height, offx, offy;
    uint64_t seed_base;
    int reseed;                      /* 1: states seeded from seed_base+idx at launch; 0: use states[] */
    orc_rng *states;                 /* optional persistent states (rendering.cu:209,232) */
    float *fb_r, *fb_g, *fb_b;       /* block-linear, grid sized, quantised 0..255 */
    float *lin_r, *lin_g, *lin_b;    /* optional: sRGB in [0,1] before expand_sRGB */
    float *xyz_x, *xyz_y, *xyz_z;    /* optional: raw XYZ sums */
    uint32_t block_lo, block_hi, block_stride; /* this call renders blocks b with b%stride==lo?  see worker */
    volatile uint32_t *next_block;
    orc_stats stats;
} render_job;

/* rendering.cu:151-235 for one row (threadIdx.y = tyi) of one block of tx*ty lanes.  Pixels are independent (own RNG stream),
 * so the unit of work a host thread takes is a block ROW: a handful of blocks at thousands of spp still keeps every thread busy. */
static void render_block_row(render_job *J, uint32_t block_idx, uint32_t tyi) {
    const uint32_t block_size = J->tx * J->ty;
    const uint32_t gbx = block_idx % J->bx, gby = block_idx / J->bx;
        for (uint32_t txi = 0; txi < J->tx; txi++) {
            uint32_t i = txi + gbx * J->tx;
            uint32_t j = tyi + gby * J->ty;
            uint32_t thread_in_block_idx = tyi * J->tx + txi;
            uint32_t idx = thread_in_block_idx + block_size * block_idx;
            if (i >= J->width || j >= J->height) continue;            /* rendering.cu:205 */
            orc_rng rs;
            if (J->reseed) orc_rng_init(J->seed_base + idx, &rs);     /* rendering.cu:137 */
            else rs = J->states[idx];                                 /* rendering.cu:209 */
            vec3 pixel_color = V(0.f, 0.f, 0.f);
            if (J->sc->valid) {
                for (uint32_t k = 0; k < J->spp; k++) {
                    ray_t r;
                    get_ray(&r, J->offx + i, J->offy + j, J->cam, &rs);
                    J->stats.paths++;
                    ray_bounce(J->sc, &r, J->bounce_limit, &rs, &J->stats);
                    pixel_color = vadd(pixel_color,
                                       dev_spectrum_to_XYZ(r.wavelengths, r.power_distr, N_RAY_WAVELENGTHS, r.valid_wavelengths));
                }
            }
            if (J->states) J->states[idx] = rs;                       /* rendering.cu:232 */
            if (J->xyz_x) { J->xyz_x[idx] = pixel_color.e[0]; J->xyz_y[idx] = pixel_color.e[1]; J->xyz_z[idx] = pixel_color.e[2]; }
            /* save_to_fb, rendering.cu:140-149 */
            vec3 lin = XYZ_to_sRGB(vdiv(pixel_color, (float)J->spp), d65_XYZ_to_sRGB);
            if (J->lin_r) { J->lin_r[idx] = lin.e[0]; J->lin_g[idx] = lin.e[1]; J->lin_b[idx] = lin.e[2]; }
            vec3 q = expand_sRGB(lin);
            J->fb_r[idx] = q.e[0]; J->fb_g[idx] = q.e[1]; J->fb_b[idx] = q.e[2];
        }
}

static void *render_worker(void *arg) {
    /* work on a stack copy: the per-ray counters must not share cache lines between threads */
    render_job local = *(render_job *)arg;
    render_job *J = &local;
    for (;;) {
        uint32_t k = __sync_fetch_and_add(J->next_block, 1u);      /* work item k = row k % ty of the (k / ty)-th selected block */
        uint32_t b = J->block_lo + (k / J->ty) * J->block_stride;
        if (b >= J->block_hi) break;
        render_block_row(J, b, k % J->ty);
    }
    ((render_job *)arg)->stats = local.stats;
    return NULL;
}

/* ------------------------------------------------------------------------------------------
 * exported scene / render API (used by tests through ctypes)
 * ---------------------------------------------------------------------------------------- */
ORC_API orc_scene *orc_scene_create(const orc_tri_in *tin, size_t n, const orc_material *mats, size_t m, const float *bg) {
    orc_scene *sc = (orc_scene *)calloc(1, sizeof(*sc));
    sc->n_tris = n;
    sc->tris = (tri_t *)calloc(n ? n : 1, sizeof(tri_t));
    sc->list = (tri_t **)calloc(n ? n : 1, sizeof(tri_t *));
    for (size_t k = 0; k < n; k++) {
        tri_t *t = &sc->tris[k];
        t->v[0] = v3(tin[k].v0); t->v[1] = v3(tin[k].v1); t->v[2] = v3(tin[k].v2);
        t->mat_index = tin[k].mat_index;
        t->aa_plane = (int)tin[k].aa_plane;
        tri_init(t);
        sc->list[k] = t;
    }
    sc->n_mats = m;
    sc->mats = (orc_material *)calloc(m ? m : 1, sizeof(orc_material));
    memcpy(sc->mats, mats, m * sizeof(orc_material));
    memcpy(sc->background, bg, sizeof(sc->background));
    sc->pool_cap = 2 * (n ? n : 1) + 8;
    sc->pool = (bvh_node *)calloc(sc->pool_cap, sizeof(bvh_node));
    return sc;
}

ORC_API void orc_scene_destroy(orc_scene *sc) {
    if (!sc) return;
    free(sc->tris); free(sc->list); free(sc->mats); free(sc->pool); free(sc);
}

/* scene/scene.cu:9-20 + bvh.cuh:127-135: fresh XORWOW(seed), build, boxes. */
ORC_API int orc_scene_build_bvh_reference(orc_scene *sc, uint64_t seed) {
    sc->pool_used = 0; sc->valid = 0; sc->root = NULL;
    if (sc->n_tris == 0) return 0;
    orc_rng rs;
    orc_rng_init(seed, &rs);
    if (!build_bvh(sc, &rs)) return 0;
    build_nodes_bboxes(sc->root);
    sc->valid = 1;
    return 1;
}

/* Import an externally built binary tree (the product's own builder for the big synthetic scenes):
 * node k has children left[k]/right[k] (node indices) or, when prim[k] >= 0, is a leaf for
 * triangle prim[k] (index into the ORIGINAL triangle order).  Boxes are recomputed here with the
 * reference's rule (leaf = padded tri box, internal = union), never taken from the caller. */
ORC_API int orc_scene_set_bvh(orc_scene *sc, size_t n_nodes, const int32_t *left, const int32_t *right, const int32_t *prim,
                              int32_t root) {
    sc->pool_used = 0; sc->valid = 0; sc->root = NULL;
    if (n_nodes == 0 || n_nodes > sc->pool_cap || root < 0 || (size_t)root >= n_nodes) return 0;
    for (size_t k = 0; k < n_nodes; k++) {
        bvh_node *nd = &sc->pool[k];
        nd->bbox.x = nd->bbox.y = nd->bbox.z = itv(+FLT_MAX, -FLT_MAX);
        if (prim[k] >= 0) {
            if ((size_t)prim[k] >= sc->n_tris) return 0;
            nd->is_leaf = 1; nd->left = nd->right = NULL; nd->primitive = &sc->tris[prim[k]];
        } else {
            if (left[k] < 0 || right[k] < 0 || (size_t)left[k] >= n_nodes || (size_t)right[k] >= n_nodes) return 0;
            nd->is_leaf = 0; nd->primitive = NULL;
            nd->left = &sc->pool[left[k]]; nd->right = &sc->pool[right[k]];
        }
    }
    sc->pool_used = n_nodes;
    sc->root = &sc->pool[root];
    build_nodes_bboxes(sc->root);
    sc->valid = 1;
    return 1;
}

ORC_API size_t orc_scene_node_count(const orc_scene *sc) { return sc->pool_used; }

/* Pre-order dump: for node k (pre-order rank) left/right = pre-order ranks of children or -1,
 * prim = original triangle index for leaves else -1, box = 6 floats (xmin,xmax,ymin,ymax,zmin,zmax). */
ORC_API size_t orc_scene_get_bvh(const orc_scene *sc, int32_t *left, int32_t *right, int32_t *prim, float *boxes) {
    if (!sc->valid) return 0;
    size_t count = 0, top = 0, cap = sc->pool_used + 1;
    const bvh_node **st = (const bvh_node **)malloc(cap * sizeof(*st));
    int32_t *parent = (int32_t *)malloc(cap * sizeof(int32_t));
    char *is_right = (char *)malloc(cap);
    st[top] = sc->root; parent[top] = -1; is_right[top] = 0; top++;
    while (top) {
        top--;
        const bvh_node *n = st[top];
        int32_t me = (int32_t)count++;
        if (parent[top] >= 0) { if (is_right[top]) right[parent[top]] = me; else left[parent[top]] = me; }
        left[me] = right[me] = -1;
        prim[me] = n->is_leaf ? (int32_t)(n->primitive - sc->tris) : -1;
        aabb_t b = node_bounding_box(n);
        float *o = boxes + 6 * (size_t)me;
        o[0] = b.x.min; o[1] = b.x.max; o[2] = b.y.min; o[3] = b.y.max; o[4] = b.z.min; o[5] = b.z.max;
        if (!n->is_leaf) {
            st[top] = n->right; parent[top] = me; is_right[top] = 1; top++;
            st[top] = n->left; parent[top] = me; is_right[top] = 0; top++;
        }
    }
    free(st); free(parent); free(is_right);
    return count;
}

/* Precomputed triangle record dump (for parity of the product's host-side tri::init):
 * out[k] = { n.x n.y n.z D  clockwise aa_plane  bbox(6) } as 12 floats (flags as float). */
ORC_API void orc_scene_get_tris(const orc_scene *sc, float *out) {
    for (size_t k = 0; k < sc->n_tris; k++) {
        const tri_t *t = &sc->tris[k];
        float *o = out + 12 * k;
        o[0] = t->normal.e[0]; o[1] = t->normal.e[1]; o[2] = t->normal.e[2]; o[3] = t->D;
        o[4] = (float)t->clockwise; o[5] = (float)t->aa_plane;
        o[6] = t->bbox.x.min; o[7] = t->bbox.x.max; o[8] = t->bbox.y.min; o[9] = t->bbox.y.max;
        o[10] = t->bbox.z.min; o[11] = t->bbox.z.max;
    }
}

/* One chunk of the image (render_manager.cu:3-66 -> rendering.cu:244-277), all blocks b with
 * b % block_stride == block_lo (block_stride = 1 renders everything; >1 is the multi-rank split).
 * Buffers are block-linear and grid sized (tx*bx*ty*by floats each).  states may be NULL. */
ORC_API int orc_render(const orc_scene *sc, const orc_camera_data *cam, uint32_t spp, uint32_t bounce_limit, uint32_t tx,
                       uint32_t ty, uint32_t bx, uint32_t by, uint32_t width, uint32_t height, uint32_t offx, uint32_t offy,
                       uint64_t seed_base, int reseed, orc_rng *states, uint32_t block_lo, uint32_t block_stride, float *fb_r,
                       float *fb_g, float *fb_b, float *lin_r, float *lin_g, float *lin_b, float *xyz_x, float *xyz_y,
                       float *xyz_z, int n_threads, orc_stats *stats_out) {
    if (!sc || !cam || !fb_r || !fb_g || !fb_b || tx == 0 || ty == 0 || block_stride == 0) return -1;
    if (!reseed && !states) return -2;
    /* Q17: the reference narrows these to 16 bit (rendering.cu:154,245) */
    spp = (uint16_t)spp; bounce_limit = (uint16_t)bounce_limit;
    width = (uint16_t)width; height = (uint16_t)height; offx = (uint16_t)offx; offy = (uint16_t)offy;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    volatile uint32_t next = 0;
    render_job *jobs = (render_job *)calloc((size_t)n_threads, sizeof(render_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int k = 0; k < n_threads; k++) {
        render_job *J = &jobs[k];
        J->sc = sc; J->cam = cam; J->spp = spp; J->bounce_limit = bounce_limit;
        J->tx = tx; J->ty = ty; J->bx = bx; J->by = by;
        J->width = width; J->height = height; J->offx = offx; J->offy = offy;
        J->seed_base = seed_base; J->reseed = reseed; J->states = states;
        J->fb_r = fb_r; J->fb_g = fb_g; J->fb_b = fb_b;
        J->lin_r = lin_r; J->lin_g = lin_g; J->lin_b = lin_b;
        J->xyz_x = xyz_x; J->xyz_y = xyz_y; J->xyz_z = xyz_z;
        J->block_lo = block_lo; J->block_hi = bx * by; J->block_stride = block_stride;
        J->next_block = &next;
    }
    if (n_threads == 1) render_worker(&jobs[0]);
    else {
        for (int k = 0; k < n_threads; k++) pthread_create(&th[k], NULL, render_worker, &jobs[k]);
        for (int k = 0; k < n_threads; k++) pthread_join(th[k], NULL);
    }
    if (stats_out) {
        memset(stats_out, 0, sizeof(*stats_out));
        for (int k = 0; k < n_threads; k++) {
            stats_out->rays += jobs[k].stats.rays; stats_out->paths += jobs[k].stats.paths;
            stats_out->trav_iters += jobs[k].stats.trav_iters; stats_out->box_tests += jobs[k].stats.box_tests;
            stats_out->tri_tests += jobs[k].stats.tri_tests;
            if (jobs[k].stats.max_stack > stats_out->max_stack) stats_out->max_stack = jobs[k].stats.max_stack;
        }
    }
    free(jobs); free(th);
    return 0;
}

/* render_manager.cuh:68-142: block-linear chunk buffer -> row-major image */
ORC_API void orc_unswizzle(const float *src, float *dst, uint32_t tx, uint32_t ty, uint32_t bx, uint32_t by, uint32_t n_cols,
                           uint32_t n_rows, uint32_t offs_x, uint32_t offs_y, uint32_t image_width) {
    uint32_t block_size = tx * ty, grid_size = bx * by;
    for (uint32_t idx = 0; idx < block_size * grid_size; idx++) {
        uint32_t blockIdx = idx / block_size;
        uint32_t block_x = blockIdx % bx, block_y = blockIdx / bx;
        uint32_t thread_idx = idx % block_size;
        uint32_t thread_x = thread_idx % tx, thread_y = thread_idx / tx;
        uint32_t fb_x = tx * block_x + thread_x;
        uint32_t fb_y = ty * block_y + thread_y;
        if (fb_x < n_cols && fb_y < n_rows) {
            fb_x += offs_x; fb_y += offs_y;
            dst[(size_t)fb_y * image_width + fb_x] = src[idx];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * host-side inputs of the path: camera (rendering/camera.cu:7-58) and spectra baking
 * (color/color_to_spectrum.cuh:36-40,109-121,153-219; materials/material.cuh:71-84)
 * ---------------------------------------------------------------------------------------- */
static inline float degrees_to_radians(float degrees) { return degrees * ORC_PI / 180.0f; } /* cuda_utility.cuh:40-43 */

ORC_API void orc_camera_init(int image_width, int image_height, float vfov, const float lookfrom[3], const float lookat[3],
                             const float vup[3], float defocus_angle, float focus_dist, orc_camera_data *out) {
    vec3 center = v3(lookfrom);
    float theta = degrees_to_radians(vfov);
    float h = tanf(theta / 2.0f) * focus_dist;
    float viewport_height = 2.0f * h;
    float viewport_width = viewport_height * ((float)image_width / (float)image_height);
    vec3 w = vunit(vsub(v3(lookfrom), v3(lookat)));
    vec3 u = vunit(vcross(v3(vup), w));
    vec3 v = vcross(w, u);
    vec3 viewport_u = vscale(viewport_width, u);
    vec3 viewport_v = vscale(viewport_height, vneg(v));
    vec3 pixel_delta_u = vdiv(viewport_u, (float)image_width);
    vec3 pixel_delta_v = vdiv(viewport_v, (float)image_height);
    vec3 viewport_upper_left = vsub(vsub(vsub(center, vscale(focus_dist, w)), vdiv(viewport_u, 2)), vdiv(viewport_v, 2));
    /* camera.cu:53: "0.5 * (du + dv)": the double literal narrows to float through operator*(float, vec3) */
    vec3 pixel00_loc = vadd(viewport_upper_left, vscale((float)0.5, vadd(pixel_delta_u, pixel_delta_v)));
    float defocus_radius = focus_dist * tanf(degrees_to_radians(defocus_angle / 2));
    vec3 ddu = vscale(defocus_radius, u);
    vec3 ddv = vscale(defocus_radius, v);
    out->width = (uint32_t)image_width; out->height = (uint32_t)image_height;
    for (int k = 0; k < 3; k++) {
        out->pixel_delta_u[k] = pixel_delta_u.e[k]; out->pixel_delta_v[k] = pixel_delta_v.e[k];
        out->pixel00_loc[k] = pixel00_loc.e[k]; out->camera_center[k] = center.e[k];
        out->defocus_disk_u[k] = ddu.e[k]; out->defocus_disk_v[k] = ddv.e[k];
    }
    out->defocus_angle = defocus_angle;
}

/* color_to_spectrum.cuh:36-40 */
static float sigmoid_inf_check(float x) {
    if (isinf(x)) return x > 0 ? 1 : 0;
    return 0.5f * x / sqrtf(1.0f + x * x) + 0.5f;
}
/* color_to_spectrum.cuh:153-156 */
static float polynomial(float x, float c2, float c1, float c0) { return x * x * c2 + x * c1 + c0; }

/* Sigmoid-polynomial spectrum from explicit coefficients, sampled the reference's way:
 * lambda accumulates by step = 470/95 (Q4), value = scale * sigmoid(poly) [* D65n(lambda)]
 * (color_to_spectrum.cuh:173-186, 204-219).  coeffs = (x, y, z) of the reference's vec3, i.e.
 * the evaluator uses z as the quadratic term, y linear, x constant (Q2). */
ORC_API void orc_bake_sigmoid_spectrum(const float coeffs[3], float scale, int times_d65, float *sampled_spectrum) {
    float step = (LAMBDA_MAX - LAMBDA_MIN) / N_CIE_SAMPLES;
    float lambda = LAMBDA_MIN;
    for (int i = 0; i < N_CIE_SAMPLES; i++) {
        float x = polynomial(lambda, coeffs[2], coeffs[1], coeffs[0]);
        float s = sigmoid_inf_check(x);
        float val = times_d65 ? scale * s * orc_spectrum_interp(normalized_cie_d65, lambda, N_CIE_SAMPLES) : s;
        sampled_spectrum[i] = val;
        lambda += step;
    }
}

/* Grey branch of dev_get_sigmoid_coeffs (color_to_spectrum.cuh:118-120): r==g==b needs no table. */
ORC_API int orc_grey_sigmoid_coeffs(const float rgb[3], float coeffs[3]) {
    float r = rgb[0], g = rgb[1], b = rgb[2];
    if (!(r == g && g == b)) return 0;       /* needs the rgb2spec table, absent from the reference mount */
    coeffs[0] = 0.0f; coeffs[1] = 0.0f; coeffs[2] = (r - .5f) / sqrtf(r * (1 - r));
    return 1;
}

/* material::compute_spectral_distr (material.cuh:71-84) for materials that need no table. */
ORC_API int orc_material_bake(orc_material *m) {
    float c[3];
    switch (m->material_type) {
    case MAT_EMISSIVE:
        if (!orc_grey_sigmoid_coeffs(m->col, c)) return 0;
        /* pow(power, 2.0f) (color_to_spectrum.cuh:181) */
        orc_bake_sigmoid_spectrum(c, orc_powf(m->emission_power, 2.0f), 1, m->spectral_distribution);
        return 1;
    case MAT_DIELECTRIC:
        for (int i = 0; i < N_CIE_SAMPLES; i++) m->spectral_distribution[i] = 1.0f;
        return 1;
    default:
        if (!orc_grey_sigmoid_coeffs(m->col, c)) return 0;
        orc_bake_sigmoid_spectrum(c, 1.0f, 0, m->spectral_distribution);
        return 1;
    }
}

/* Background: host srgb_to_illuminance_spectrum (color_to_spectrum.cuh:158-171, rendering.cu:324), grey only. */
ORC_API int orc_background_spectrum(const float rgb[3], float *out) {
    float c[3];
    if (!orc_grey_sigmoid_coeffs(rgb, c)) return 0;
    orc_bake_sigmoid_spectrum(c, orc_powf(1.0f, 2.0f), 1, out);
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * small KAT entry points
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_hero_wavelengths(uint64_t seed, float *out7) {
    orc_rng s; orc_rng_init(seed, &s);
    init_hero_wavelength(out7, N_RAY_WAVELENGTHS, &s);
}
ORC_API void orc_spectrum_to_XYZ(const float *wl, const float *power, uint32_t n_valid, float *out3) {
    vec3 c = dev_spectrum_to_XYZ(wl, power, N_RAY_WAVELENGTHS, n_valid);
    out3[0] = c.e[0]; out3[1] = c.e[1]; out3[2] = c.e[2];
}
ORC_API void orc_XYZ_to_sRGB(const float *xyz, float *lin3, float *q3) {
    vec3 l = XYZ_to_sRGB(v3(xyz), d65_XYZ_to_sRGB);
    vec3 q = expand_sRGB(l);
    for (int k = 0; k < 3; k++) { lin3[k] = l.e[k]; q3[k] = q.e[k]; }
}
ORC_API void orc_refract(const float *uv, const float *n, float eta, float *out3) {
    vec3 r = vrefract(v3(uv), v3(n), eta);
    out3[0] = r.e[0]; out3[1] = r.e[1]; out3[2] = r.e[2];
}
ORC_API void orc_reflect(const float *v, const float *n, float *out3) {
    vec3 r = vreflect(v3(v), v3(n));
    out3[0] = r.e[0]; out3[1] = r.e[1]; out3[2] = r.e[2];
}
ORC_API void orc_unit_vector(const float *v, float *out3) {
    vec3 r = vunit(v3(v));
    out3[0] = r.e[0]; out3[1] = r.e[1]; out3[2] = r.e[2];
}
ORC_API float orc_cie_table(int which, int k) {
    const float *t = which == 0 ? cie_x : which == 1 ? cie_y : which == 2 ? cie_z : normalized_cie_d65;
    return t[k];
}
ORC_API void orc_color_matrix(float out9[9]) { memcpy(out9, d65_XYZ_to_sRGB, sizeof(d65_XYZ_to_sRGB)); }   /* utils/color_const.cu:17-19 */
ORC_API float orc_cie_interp(int which, float lambda) {
    const float *t = which == 0 ? cie_x : which == 1 ? cie_y : which == 2 ? cie_z : normalized_cie_d65;
    return orc_spectrum_interp(t, lambda, N_CIE_SAMPLES);
}
/* closest hit of one explicit ray against the scene: returns hit flag; out = t, p(3), normal(3), front, mat */
ORC_API int orc_trace_ray(const orc_scene *sc, const float *o, const float *d, float *out9) {
    ray_t r; memset(&r, 0, sizeof(r));
    r.orig = v3(o); r.dir = v3(d);
    hit_record rec; memset(&rec, 0, sizeof(rec));
    orc_stats st; memset(&st, 0, sizeof(st));
    int h = bvh_hit(&r, 0.0f, FLT_MAX, &rec, sc->root, &st);
    if (h) {
        out9[0] = rec.t;
        for (int k = 0; k < 3; k++) { out9[1 + k] = rec.p.e[k]; out9[4 + k] = rec.normal.e[k]; }
        out9[7] = (float)rec.front_face; out9[8] = (float)rec.mat_index;
    }
    return h;
}
/* one scatter event with an explicit RNG seed: in/out ray fields; returns did_scatter */
ORC_API int orc_scatter(const orc_material *mat, float *orig, float *dir, float *wavelengths, float *power, uint32_t *valid,
                        const float *p, const float *normal, float t, int front_face, uint64_t seed, uint32_t *draws_out) {
    ray_t r; memset(&r, 0, sizeof(r));
    r.orig = v3(orig); r.dir = v3(dir); r.valid_wavelengths = *valid;
    memcpy(r.wavelengths, wavelengths, sizeof(r.wavelengths));
    memcpy(r.power_distr, power, sizeof(r.power_distr));
    hit_record rec; rec.p = v3(p); rec.normal = v3(normal); rec.t = t; rec.front_face = front_face; rec.mat_index = 0;
    orc_rng s, s2; orc_rng_init(seed, &s); s2 = s;
    int did = material_scatter(mat, &r, &rec, &s);
    for (int k = 0; k < 3; k++) { orig[k] = r.orig.e[k]; dir[k] = r.dir.e[k]; }
    memcpy(power, r.power_distr, sizeof(r.power_distr));
    *valid = r.valid_wavelengths;
    if (draws_out) { uint32_t n = 0; while (memcmp(&s, &s2, sizeof(s)) != 0 && n < 100000) { orc_rng_next(&s2); n++; } *draws_out = n; }
    return did;
}
ORC_API size_t orc_sizeof(int what) {
    switch (what) {
    case 0: return sizeof(orc_material);
    case 1: return sizeof(orc_camera_data);
    case 2: return sizeof(orc_tri_in);
    case 3: return sizeof(orc_rng);
    case 4: return sizeof(orc_stats);
    default: return 0;
    }
}
