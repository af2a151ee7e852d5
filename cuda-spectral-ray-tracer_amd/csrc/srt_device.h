// srt_device.h -- gfx950 device arithmetic of the spectral path tracer.
//
// Every function restates one piece of the reference's device library (file:line cited) with the
// operation order of the reference's SOURCE under plain IEEE-754 fp32 semantics: this translation unit
// is compiled with -ffp-contract=off and correctly rounded fp32 divide/sqrt, no fast-math, and the parity
// tests require bit-identical results to the CPU oracle, because the estimator's discrete
// decisions (hit/miss at an edge, reflect/refract, rejection accept) flip on 1-ulp differences.
// "Exact" therefore means: equal to the no-FMA reading of the source that the CPU oracle embodies.  The real
// nvcc build contracts a*b+c into FMAs by default and has its own powf; equivalence to THAT binary is
// statistical only and cannot be pinned here (DESIGN.md sections 2 and 3).
//
// Data layout is this build's own (DESIGN.md "HBM layout"): paired-child 64-byte BVH records,
// 48-byte triangle records with the 2-D projected vertices, spectra as (s[k], s[k+1]) pairs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "srt_powf.h"

namespace srt {

constexpr int kCieSamples = 95;
constexpr int kWavelengths = 7;
constexpr float kLambdaMin = 360.0f;
constexpr float kLambdaMax = 830.0f;
constexpr float kEpsilon = 0.0001f;      // materials/material.cuh:14
constexpr float kFltMax = 3.402823466e+38f;

// one out-of-line copy of srt_powf per kernel image (it is called from four places)
__device__ __noinline__ static float dev_powf(float x, float y) { return srt_powf(x, y); }

// ------------------------------------------------------------------------------------------------
// vec3 (math/vec3.cuh:119-163).  dot is left-associated x+y+z (:149-153).
// ------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(float t, V3 v) { return mk(t * v.x, t * v.y, t * v.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float length_squared(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }   // :66-72
// unit_vector = v / len = (1/len) * v  (:144-147,:161)
__device__ __forceinline__ V3 unit_vector(V3 v) { float k = 1.0f / sqrtf(length_squared(v)); return k * v; }
// reflect = v - 2*dot(v,n)*n  (:180-183)
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (2.0f * dot(v, n)) * n; }
// refract (:199-205)
__device__ __forceinline__ V3 refract(V3 uv, V3 n, float etai_over_etat) {
    float cos_theta = fminf(dot(-uv, n), 1.0f);
    V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    V3 r_out_parallel = (-sqrtf(fabsf(1.0f - length_squared(r_out_perp)))) * n;
    return r_out_perp + r_out_parallel;
}
// IEEE-754-2019 minimum (v_minimum3_f32 on gfx950): a NaN operand makes the result NaN, -0 < +0.  `a >= 0 && b >= 0 && c >= 0`
// is `minimum(minimum(a, b), c) >= 0` for every operand (a NaN fails both forms, zeros of either sign pass both): two instructions
// and one compare instead of three compares and two mask operations (trav_fringe_compute's inside test).

// The two halves of that test as the FRINGE visit uses them (`sign` = 0 for a clockwise triangle, 0x80000000 for a
// counter-clockwise one: `a <= 0` is `-a >= 0`, a flip of the sign bit); op-sweep kinds 20 / 21 run exactly these two functions
// against the three compares of is_interior_faster (primitives/tri.cu:121-128).
__device__ __forceinline__ float inside_min2(float a1, float a2, uint32_t sign) {
    return __builtin_elementwise_minimum(__uint_as_float(__float_as_uint(a1) ^ sign), __uint_as_float(__float_as_uint(a2) ^ sign));
}
__device__ __forceinline__ bool inside_min3(float m12, float a3, uint32_t sign) {
    return __builtin_elementwise_minimum(m12, __uint_as_float(__float_as_uint(a3) ^ sign)) >= 0.f;
}

__device__ __forceinline__ bool near_zero(V3 v) {   // :93-98
    const float s = 1e-8f;
    return (fabsf(v.x) < s) && (fabsf(v.y) < s) && (fabsf(v.z) < s);
}

// ------------------------------------------------------------------------------------------------
// RNG: cuRAND XORWOW core (curand_init with subsequence 0 / offset 0, curand, curand_uniform),
// restated from the published definition; call sites rendering/rendering.cu:137,
// utils/cuda_utility.cu:19-49.  Six 32-bit words per lane.
// ------------------------------------------------------------------------------------------------
struct Rng { uint32_t d, v0, v1, v2, v3, v4; };
#ifndef SRT_XOR3
#define SRT_XOR3 1               /* three of the four operands of the XORWOW step's xor in one v_bitop3_b32 (0: the compiler's three v_xor_b32) */
#endif
#ifndef SRT_ASM_RNG
#define SRT_ASM_RNG 1            /* the two rejection loops (unit sphere, unit disk) as hand-scheduled assembly (0: the C++ loops) */
#endif

__device__ __forceinline__ void rng_seed(Rng &s, uint64_t seed) {
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    uint32_t s1 = ((uint32_t)(seed >> 32)) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    s.d = 6615241u + t1 + t0;
    s.v0 = 123456789u + t0;
    s.v1 = 362436069u ^ t0;
    s.v2 = 521288629u + t1;
    s.v3 = 88675123u ^ t1;
    s.v4 = 5783321u + t0;
}
__device__ __forceinline__ uint32_t rng_next(Rng &s) {
    uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4;
#if SRT_XOR3
    s.v4 = __builtin_amdgcn_bitop3_b32(s.v4, s.v4 << 4, t, 0x96) ^ (t << 1);      // the same four-way xor, three of its operands in one v_bitop3_b32 (0x96 = a ^ b ^ c)
#else
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
#endif
    s.d += 362437u;
    return s.v4 + s.d;
}
// cuda_random_float(): curand_uniform in (0,1]  (utils/cuda_utility.cu:19-26): fl(fl(v * 2^-32) + 2^-33) with v = (float)r.
// v * 2^-32 is exact (a power-of-two scaling of an fp32 value in [0, 2^32], no underflow), so ONE fused multiply-add
// rounds exactly where the reference's add rounds: same bits for every r (checked on the device: op sweep, op 18).
__device__ __forceinline__ float rng_unit_from_bits(uint32_t r) {
    return __builtin_fmaf((float)r, 2.3283064e-10f, 2.3283064e-10f / 2.0f);
}
__device__ __forceinline__ float rng_uniform(Rng &s) { return rng_unit_from_bits(rng_next(s)); }
// cuda_random_float(min,max): u*(max-min)+min  (utils/cuda_utility.cu:28-41)
__device__ __forceinline__ float rng_range(Rng &s, float mn, float mx) {
    float range_width = mx - mn;
    float random = rng_uniform(s);
    return random * range_width + mn;
}
// cuda_random_float(-1, 1) = fl(fl(u * 2) + -1): u * 2 is exact, and 2 * fl(v * 2^-32 + 2^-33) = fl(v * 2^-31 + 2^-32)
// (scaling by two commutes with rounding, nothing here is near the denormal range), again one fused operation with an
// exact product; the final add is the reference's (op sweep, op 19).
__device__ __forceinline__ float rng_pm1_from_bits(uint32_t r) {
    return __builtin_fmaf((float)r, 2.0f * 2.3283064e-10f, 2.3283064e-10f) + -1.0f;
}
__device__ __forceinline__ float rng_pm1(Rng &s) { return rng_pm1_from_bits(rng_next(s)); }
// Three (two) consecutive rng_pm1 draws as one block of gfx950 assembly: the rejection loops of random_in_unit_sphere / random_in_unit_disk
// run ~5.7 / ~2.9 times per shading pass for the whole wave, and the compiler rotates the five xorshift words through four (five) register
// copies per iteration.  Here the state stays in its registers -- the words that only move take two (three) copies, the new ones are
// computed in place -- and three operands of each step's four-way xor meet in one v_bitop3_b32.  Instruction for instruction rng_next +
// rng_pm1_from_bits: t = x ^ (x >> 2); v' = v ^ (v << 4) ^ t ^ (t << 1); d += 362437; r = v' + d; fl(fl((float)r * 2^-31 + 2^-32) + -1).
#define SRT_RNG_T(T, X)            "v_lshrrev_b32 " T ", 2, " X "\n\t" "v_xor_b32 " T ", " T ", " X "\n\t"
#define SRT_RNG_STEP(OUT, PREV, T, A) "v_lshlrev_b32 " A ", 4, " PREV "\n\t" "v_bitop3_b32 " A ", " PREV ", " A ", " T " bitop3:0x96\n\t" \
                                   "v_lshlrev_b32 " T ", 1, " T "\n\t" "v_xor_b32 " OUT ", " A ", " T "\n\t"
#define SRT_RNG_PM1(R, N, K)       "v_add3_u32 " R ", %[d], " N ", " K "\n\t" "v_cvt_f32_u32 " R ", " R "\n\t" \
                                   "v_fmamk_f32 " R ", " R ", 0x30000000, %[c32]\n\t" "v_add_f32 " R ", -1.0, " R "\n\t"
// The whole rejection loop: `do { p = 3 (2) draws } while (!(|p|^2 < 1))`.  EXEC is saved once, the compare that accepts a lane's point takes the lane out
// (v_cmpx), the loop runs while any lane is left; accepted lanes keep their point and their stream position because nothing writes them any more.
// |p|^2 = (x*x + y*y) + z*z as length_squared (vec3.cuh:59-61) / (x*x + y*y) + 0*0 for the disk (vec3.cuh:240-246; the + 0 cannot change a sum of squares).
// s[80:81] and VCC are scratch.
__device__ __forceinline__ void rng_sphere_loop_asm(Rng &s, float &x, float &y, float &z, float &len2) {
    uint32_t t2, t3, a;
    float t1;
    asm volatile(
        "s_mov_b64 s[80:81], exec\n\t"
        ".Lsrt_sphere_try%=:\n\t"
        SRT_RNG_T("%[t1]", "%[v0]") SRT_RNG_T("%[t2]", "%[v1]") SRT_RNG_T("%[t3]", "%[v2]")
        "v_mov_b32 %[v0], %[v3]\n\t"
        SRT_RNG_STEP("%[v2]", "%[v4]", "%[t1]", "%[a]")                 /* n1 -> v2 (its old value lives on in t3) */
        "v_mov_b32 %[v1], %[v4]\n\t"
        SRT_RNG_PM1("%[x]", "%[v2]", "%[k1]")
        SRT_RNG_STEP("%[v3]", "%[v2]", "%[t2]", "%[a]")                 /* n2 -> v3 */
        SRT_RNG_PM1("%[y]", "%[v3]", "%[k2]")
        SRT_RNG_STEP("%[v4]", "%[v3]", "%[t3]", "%[a]")                 /* n3 -> v4 */
        SRT_RNG_PM1("%[z]", "%[v4]", "%[k3]")
        "v_add_u32 %[d], %[k3], %[d]\n\t"
        "v_mul_f32 %[t1], %[x], %[x]\n\t"
        "v_mul_f32 %[t2], %[y], %[y]\n\t"
        "v_mul_f32 %[t3], %[z], %[z]\n\t"
        "v_add_f32 %[t1], %[t1], %[t2]\n\t"
        "v_add_f32 %[t1], %[t1], %[t3]\n\t"
        "v_cmpx_ngt_f32 vcc, 1.0, %[t1]\n\t"                          /* stay while !(|p|^2 < 1) */
        "s_cbranch_execnz .Lsrt_sphere_try%=\n\t"
        "s_mov_b64 exec, s[80:81]\n\t"
        : [v0] "+v"(s.v0), [v1] "+v"(s.v1), [v2] "+v"(s.v2), [v3] "+v"(s.v3), [v4] "+v"(s.v4), [d] "+v"(s.d),
          [x] "=&v"(x), [y] "=&v"(y), [z] "=&v"(z), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [a] "=&v"(a)
        : [k1] "s"(362437u), [k2] "s"(2u * 362437u), [k3] "s"(3u * 362437u), [c32] "v"(2.3283064e-10f)
        : "s80", "s81", "vcc");
    len2 = t1;      // (x*x + y*y) + z*z of the accepted point: what unit_vector takes the root of
}
__device__ __forceinline__ void rng_disk_loop_asm(Rng &s, float &x, float &y) {
    uint32_t t1, t2, a;
    asm volatile(
        "s_mov_b64 s[80:81], exec\n\t"
        ".Lsrt_disk_try%=:\n\t"
        SRT_RNG_T("%[t1]", "%[v0]") SRT_RNG_T("%[t2]", "%[v1]")
        "v_mov_b32 %[v0], %[v2]\n\t"
        "v_mov_b32 %[v1], %[v3]\n\t"
        SRT_RNG_STEP("%[v3]", "%[v4]", "%[t1]", "%[a]")                 /* n1 -> v3 */
        "v_mov_b32 %[v2], %[v4]\n\t"
        SRT_RNG_PM1("%[x]", "%[v3]", "%[k1]")
        SRT_RNG_STEP("%[v4]", "%[v3]", "%[t2]", "%[a]")                 /* n2 -> v4 */
        SRT_RNG_PM1("%[y]", "%[v4]", "%[k2]")
        "v_add_u32 %[d], %[k2], %[d]\n\t"
        "v_mul_f32 %[t1], %[x], %[x]\n\t"
        "v_mul_f32 %[t2], %[y], %[y]\n\t"
        "v_add_f32 %[t1], %[t1], %[t2]\n\t"
        "v_cmpx_ngt_f32 vcc, 1.0, %[t1]\n\t"
        "s_cbranch_execnz .Lsrt_disk_try%=\n\t"
        "s_mov_b64 exec, s[80:81]\n\t"
        : [v0] "+v"(s.v0), [v1] "+v"(s.v1), [v2] "+v"(s.v2), [v3] "+v"(s.v3), [v4] "+v"(s.v4), [d] "+v"(s.d),
          [x] "=&v"(x), [y] "=&v"(y), [t1] "=&v"(t1), [t2] "=&v"(t2), [a] "=&v"(a)
        : [k1] "s"(362437u), [k2] "s"(2u * 362437u), [c32] "v"(2.3283064e-10f)
        : "s80", "s81", "vcc");
}
// random_in_unit_sphere (math/vec3.cuh:210-218); draws x, y, z in that order (DESIGN.md D1 / SURVEY Q19)
__device__ __forceinline__ V3 random_in_unit_sphere(Rng &s) {
#if SRT_ASM_RNG
    {
        float x, y, z, len2;
        rng_sphere_loop_asm(s, x, y, z, len2);
        return mk(x, y, z);
    }
#endif
    for (;;) {
        float x = rng_pm1(s);
        float y = rng_pm1(s);
        float z = rng_pm1(s);
        V3 p = mk(x, y, z);
        if (length_squared(p) < 1.0f) return p;
    }
}

// random_unit_vector = unit_vector(random_in_unit_sphere()) (vec3.cuh:221-227): the accepted point's |p|^2 is the one the loop's
// test computed (same three products, same two sums), so the assembly loop hands it over instead of the caller squaring again
__device__ __forceinline__ V3 random_unit_vector(Rng &s) {
#if SRT_ASM_RNG
    float x, y, z, len2;
    rng_sphere_loop_asm(s, x, y, z, len2);
    const float k = 1.0f / sqrtf(len2);
    return k * mk(x, y, z);
#else
    return unit_vector(random_in_unit_sphere(s));
#endif
}

// ------------------------------------------------------------------------------------------------
// spectrum (spectrum/spectrum.cu:11-48)
// ------------------------------------------------------------------------------------------------
// spectrum_interp's index/weight part (:11-20): x = (lambda-360)*(94/470); off = clamp((int)x, 0, 93); w = x - off
__device__ __forceinline__ void interp_coords(float lambda, int &offset, float &weight) {
    lambda -= kLambdaMin;
    lambda *= ((float)kCieSamples - 1) / (kLambdaMax - kLambdaMin);
    offset = (int)lambda;
    if (offset < 0) offset = 0;
    if (offset > kCieSamples - 2) offset = kCieSamples - 2;
    weight = lambda - (float)offset;
}
// (:21) on a stored (s[k], s[k+1]) pair
__device__ __forceinline__ float interp_pair(float2 pr, float weight) { return (1.0f - weight) * pr.x + weight * pr.y; }

// init_hero_wavelength (:31-48): hero = u*470+360; next = prev + 470/7 wrapped into [360,830].  The six rotations are a
// pure function of the hero, so a lane keeps only the hero and re-derives the set (same additions, same bits).
__device__ __forceinline__ float hero_draw(Rng &s) { return rng_range(s, kLambdaMin, kLambdaMax); }
__device__ __forceinline__ void hero_expand(float hero, float (&wl)[kWavelengths]) {
    const float step = (kLambdaMax - kLambdaMin) / (float)kWavelengths;
    wl[0] = hero;
    float lambda = hero;
#pragma unroll
    for (int i = 1; i < kWavelengths; i++) {
        lambda += step;
        if (lambda > kLambdaMax) {
            float remainder = lambda - kLambdaMax;
            lambda = kLambdaMin + remainder;
        }
        wl[i] = lambda;
    }
}

// ------------------------------------------------------------------------------------------------
// colour (color/color.cu:15-49)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float correct_channel(float value) {   // :15-22
    return value < 0.0f ? 0.0f
         : (value < 0.0031308f ? 12.92f * value
         : (value < 1.0f ? ((1.055f * dev_powf(value, 0.416666f)) - 0.055f) : 1.0f));
}

// ------------------------------------------------------------------------------------------------
// refraction / materials (refraction/sellmeier.cu:11-22, materials/material.cu:39-53)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sellmeier_index(float b0, float b1, float b2, float c0, float c1, float c2, float lambda) {
    lambda *= 1e-3f;
    float l2 = lambda * lambda;
    float index = 1.0f + (b0 * l2) / (l2 - c0) + (b1 * l2) / (l2 - c1) + (b2 * l2) / (l2 - c2);
    return sqrtf(index);
}
__device__ __forceinline__ float reflectance(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * srt_pow5f(1.0f - cosine);   // pow(1 - cosine, 5.0f), material.cu:48 (DESIGN D3)
}

// ------------------------------------------------------------------------------------------------
// Intersection tests
// ------------------------------------------------------------------------------------------------
// Triangle record (3 x float4):
//   a = { n.x, n.y, n.z, D }                      tri.cuh:93-94 (normal, D)
//   b = { v0[w], v0[h], v1[w], v1[h] }            vertices projected on the aa_plane axes (tri.cu:153-176)
//   c = { v2[w], v2[h], bits(flags), 0 }          flags: bit0 = w axis is y, bit1 = h axis is z,
//                                                 bit2 = clockwise, bits 8.. = mat_index
constexpr uint32_t kTriWIsY = 1u, kTriHIsZ = 2u, kTriClockwise = 4u;

// tri::hit (primitives/tri.cu:3-45) without the record write-back; returns t through `t_out`.
__device__ __forceinline__ bool tri_test(const float4 *__restrict__ tris, int tri, V3 o, V3 d, float tmin, float tmax,
                                         float &t_out) {
    const float4 a = tris[3 * tri + 0];
    const V3 n = mk(a.x, a.y, a.z);
    float denom = dot(n, d);
    if (fabsf(denom) < 1e-8f) return false;
    float t = (a.w - dot(n, o)) / denom;
    if (!(tmin <= t && t <= tmax)) return false;            // interval::contains, math/interval.cuh:44-46
    const float4 b = tris[3 * tri + 1];
    const float4 c = tris[3 * tri + 2];
    const uint32_t flags = __float_as_uint(c.z);
    // intersection = orig + t*dir (ray.cuh:31-34), only the two projected components are needed here
    const bool wy = flags & kTriWIsY, hz = flags & kTriHIsZ;
    float pw = (wy ? o.y : o.x) + t * (wy ? d.y : d.x);
    float ph = (hz ? o.z : o.y) + t * (hz ? d.z : d.y);
    // is_interior_faster (tri.cu:121-128) with double_signed_area_2D(v1,v2,v3) =
    //   (v1[w]-v3[w])*(v2[h]-v3[h]) - (v2[w]-v3[w])*(v1[h]-v3[h])   (tri.cu:181)
    float a1 = (pw - b.z) * (b.y - b.w) - (b.x - b.z) * (ph - b.w);   // (p, v0, v1)
    float a2 = (pw - c.x) * (b.w - c.y) - (b.z - c.x) * (ph - c.y);   // (p, v1, v2)
    float a3 = (pw - b.x) * (c.y - b.y) - (c.x - b.x) * (ph - b.y);   // (p, v2, v0)
    bool inside = (flags & kTriClockwise) ? (a1 >= 0.f && a2 >= 0.f && a3 >= 0.f) : (a1 <= 0.f && a2 <= 0.f && a3 <= 0.f);
    if (!inside) return false;
    t_out = t;
    return true;
}

// BVH paired-child record (4 x float4 = 64 B, one per INTERNAL node of the reference's binary tree):
//   q0 = { L.xmin, L.xmax, L.ymin, L.ymax }   q1 = { L.zmin, L.zmax, R.xmin, R.xmax }
//   q2 = { R.ymin, R.ymax, R.zmin, R.zmax }   q3 = { bits(lref), bits(rref), 0, 0 }
// ref >= 0: index of the child's own record (internal child; its box is the one stored here);
// ref <  0: ~ref is a triangle index (leaf child; leaves are tested directly, bvh.cu:73-76, no box).

// bvh::hit (bvh/bvh.cu:98-166) as a RESUMABLE per-lane state: closest hit, left child then right child with the
// updated closest distance, "t <= closest" accepts (Q11), push right iff both internal children are hit.
// The state lives in registers (node, sp, closest, hit) plus this lane's column of the LDS stack (element k at
// stack[k*64]), so a wave can interleave traversal steps of some lanes with shading of others.
struct TravStats {
    uint32_t n_iters = 0, n_tri = 0, n_box = 0;   // per lane: node records visited, triangle / box tests
    uint32_t n_nan = 0;                            // per lane: queries answered without traversal (NaN direction)
    uint32_t w_iters = 0;                          // wave: traversal steps executed (every lane counts the same)
    uint32_t w_alive = 0;                          // wave: sum over those steps of lanes that still own work
    uint32_t w_fringe = 0, l_fringe = 0, l_inner = 0;   // wave: fringe steps, lanes served by fringe / inner steps
};

struct Trav {
    int node;        // >= 0: record to visit next; kTravDone (-1): query finished, result not yet shaded; kTravIdle (-2): no query
    uint32_t sp;     // LDS byte address of slot `entries in use` of this lane's stack column (the address IS the counter: no multiply, no add)
    int top;         // the newest stack entry lives in a register; -1 when the stack is empty.  LDS holds entries 0 .. sp-2
    float c;         // closest_so_far
    int hit;         // triangle of the closest hit or -1
    uint32_t nf[3];  // LDS byte address of the near pair of inner record 0, per axis (see NodeSrc)
};
constexpr int kTravDone = -1, kTravIdle = -2;

// Start a closest-hit query (bvh.cu:101-119).  Returns true when the query is already finished (leaf root).
template <bool COUNT>
__device__ __forceinline__ bool trav_begin(Trav &tv, uint32_t stack_base, const float4 *__restrict__ tris, int root_ref, V3 o, V3 d, TravStats &ts) {
    tv.c = kFltMax; tv.hit = -1; tv.sp = stack_base; tv.top = -1;
    // A direction with a NaN component can never hit anything: every product with it is NaN (0*NaN included), so
    // dot(n, dir) is NaN, `fabs(denom) < 1e-8` is false, t = x/NaN is NaN and `0 <= t` is false for EVERY triangle
    // (tri.cu:12-23), while every box test passes (all comparisons false, aabb.cu:30-36).  The reference therefore
    // walks the whole tree and returns "miss" (NaN IOR from the Sellmeier quirk Q1); the result is known
    // without walking.  The instrumented build records the skipped query in ts.n_nan.
    if (d.x != d.x || d.y != d.y || d.z != d.z) {
        if (COUNT) ts.n_nan++;
        tv.node = kTravDone;
        return true;
    }
    if (root_ref < 0) {   // root is a leaf: only one element
        float t;
        if (COUNT) ts.n_tri++;
        if (tri_test(tris, ~root_ref, o, d, 0.0f, tv.c, t)) { tv.c = t; tv.hit = ~root_ref; }
        tv.node = kTravDone;
        return true;
    }
    tv.node = root_ref;
    return false;
}

// Two fp32 values per lane; + and * on it compile to v_pk_add_f32 / v_pk_mul_f32, which round each half exactly like the
// scalar instructions (checked on the device by the op sweep, ops 14-16).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }

// One iteration of the do-while of bvh.cu:120-162 for a lane with tv.node >= 0.
//
// The reference handles the two children one after the other (leaf -> tri::hit, internal -> aabb::hit), which on a
// wave means four divergent code segments per visit.  Everything except the comparisons against closest_so_far is
// independent of it, so this version computes the c-independent parts of BOTH children together -- both slab tests
// with packed fp32, both triangle tests in one merged segment -- and then applies the c-dependent decisions in the
// reference's order (left, then right with the updated closest):
//   box  : pass  <=>  !(min(c, t1x, t1y, t1z) <= max(tmin, t0x, t0y, t0z))     (aabb.cu:30-36)
//   tri  : hit   <=>  !(|denom| < 1e-8) && tmin <= t && t <= c && inside        (tri.cu:12-28)
// Same expressions, same operand order per value; only the instruction schedule differs.
// Paired-child record fetch: the first n_cached records (top of the tree, breadth-first) live in LDS.  The LDS pointer
// keeps its address space so that the two sides stay ds_read_b128 / global_load_dwordx4 (a generic pointer would let
// the compiler merge them into one flat_load through a selected address).
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f4v lds_cf4;
__device__ __forceinline__ f4v as_f4v(u4v v) { return __builtin_bit_cast(f4v, v); }

// Global scene arrays are read through buffer descriptors: buffer_load_dwordx4 with a 32-bit byte offset.  The
// intrinsic keeps every load a full 16-byte access (plain loads get re-shaped into dwordx3 / dwordx2 pieces when a
// component is unused) and the hardware range check turns an out-of-range index into zeros instead of a fault.
typedef __amdgpu_buffer_rsrc_t buf_rsrc;
__device__ __forceinline__ buf_rsrc make_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f4v buf_load16(buf_rsrc r, uint32_t byte_offset) {
    return as_f4v(__builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_offset, 0, 0));
}
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 buf_load8(buf_rsrc r, uint32_t byte_offset) {
    const u2v v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_offset, 0, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}

// Node record, 64 B: three axis planes (lo_L, lo_R, hi_L, hi_R) for x, y, z -- the boxes of the left / right child side
// by side, so that one packed operation serves both children -- then lref, rref (child references: >= 0 record index,
// < 0 ~triangle) and two pad words.  The LDS image of the INNER records (both children internal) keeps the three planes
// as separate float4 arrays plus one or two planes of child references: 52 B per record when all record indices fit
// 16 bit (56 B otherwise), so that a whole ~2 300-record inner tree (cfg 3) is resident and an INNER step never leaves
// the CU.
//
// aabb::hit swaps t0 / t1 when 1/dir is negative (aabb.cu:21-25); selecting the operands before the arithmetic is the
// same thing.  In LDS the selection is free: the near pair of an axis is an 8-byte read at +0 (dir >= 0) or +8 inside
// the plane entry and the far pair is the other half, so a ray keeps three byte addresses (nf) and the step issues six
// ds_read_b64 instead of three ds_read_b128 plus a dozen v_cndmask.
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef __attribute__((address_space(3))) const f2 lds_cf2;
typedef __attribute__((address_space(3))) const char lds_cchar;
struct NodeSrc {
    buf_rsrc global_nodes;               // INNER records, 64 B each
    buf_rsrc global_nodes_sw;            // the same records pre-swizzled, 80 B each (flatten_scene): read by inner_burst4_mixed_asm
    buf_rsrc global_fringe;              // FRINGE records, 96 B each (fringe_stride apart), indexed by record - n_inner
    uint32_t fringe_stride;
    int n_inner;
    lds_cf4 *lds_q0, *lds_q1, *lds_q2;   // x / y / z planes
    lds_cu32 *lds_r0, *lds_r1;           // NARROW: r0 = lref | rref << 16; else r0 = lref, r1 = rref
    int n_cached;
};
struct BoxPairs { f2 nx, fx, ny, fy, nz, fz; };   // near / far plane of each axis, .x = left child, .y = right child

__device__ __forceinline__ void select_near_far(const f4v &q, bool positive, f2 &nr, f2 &fr) {
    nr = positive ? mk2(q.x, q.y) : mk2(q.z, q.w);
    fr = positive ? mk2(q.z, q.w) : mk2(q.x, q.y);
}
__device__ __forceinline__ void fetch_node_global(const NodeSrc &ns, int node, V3 inv, BoxPairs &b, int &lref, int &rref) {
    const uint32_t off = (uint32_t)node * 64u;
    const f4v q0 = buf_load16(ns.global_nodes, off), q1 = buf_load16(ns.global_nodes, off + 16u);
    const f4v q2 = buf_load16(ns.global_nodes, off + 32u), q3 = buf_load16(ns.global_nodes, off + 48u);
    select_near_far(q0, inv.x >= 0, b.nx, b.fx);
    select_near_far(q1, inv.y >= 0, b.ny, b.fy);
    select_near_far(q2, inv.z >= 0, b.nz, b.fz);
    lref = (int)__float_as_uint(q3.x); rref = (int)__float_as_uint(q3.y);
}
// LDS byte address of the near pair of record 0 on every axis, for this ray
__device__ __forceinline__ void ray_near_addresses(const NodeSrc &ns, V3 inv, uint32_t nf[3]) {
    nf[0] = (uint32_t)(uintptr_t)ns.lds_q0 + (inv.x >= 0 ? 0u : 8u);
    nf[1] = (uint32_t)(uintptr_t)ns.lds_q1 + (inv.y >= 0 ? 0u : 8u);
    nf[2] = (uint32_t)(uintptr_t)ns.lds_q2 + (inv.z >= 0 ? 0u : 8u);
}
// record that may be LDS resident (inner records only)
template <bool NARROW>
__device__ __forceinline__ void fetch_node(const NodeSrc &ns, int node, V3 inv, const uint32_t nf[3], BoxPairs &b, int &lref, int &rref) {
    if (node < ns.n_cached) {
        const uint32_t rec = (uint32_t)node << 4;
        const uint32_t ax = rec + nf[0], ay = rec + nf[1], az = rec + nf[2];
        b.nx = *(lds_cf2 *)(uintptr_t)ax; b.fx = *(lds_cf2 *)(uintptr_t)(ax ^ 8u);
        b.ny = *(lds_cf2 *)(uintptr_t)ay; b.fy = *(lds_cf2 *)(uintptr_t)(ay ^ 8u);
        b.nz = *(lds_cf2 *)(uintptr_t)az; b.fz = *(lds_cf2 *)(uintptr_t)(az ^ 8u);
        const uint32_t r0 = ns.lds_r0[node];
        if (NARROW) { lref = (int)(r0 & 0xffffu); rref = (int)(r0 >> 16); }
        else { lref = (int)r0; rref = (int)ns.lds_r1[node]; }
    } else {
        fetch_node_global(ns, node, inv, b, lref, rref);
    }
}

// Traversal stack: this lane's column of LDS slots, slot k at index k*64; 16-bit slots when every record index fits 15 bits
// (NARROW).  Slots 0 and 1 are sentinels holding -1, entry k of the stack lives in slot k + 2.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) int32_t lds_i32;
typedef __attribute__((address_space(3))) int16_t lds_i16;
constexpr int kStackSentinels = 2;
struct StackRef {
    lds_i32 *s32;      // used when !NARROW
    lds_i16 *s16;      // used when NARROW
};
template <bool NARROW>
__device__ __forceinline__ void stack_init(const StackRef &st) {
    if (NARROW) { st.s16[0] = (int16_t)-1; st.s16[64] = (int16_t)-1; } else { st.s32[0] = -1; st.s32[64] = -1; }
}

// aabb::hit (bvh/aabb.cu:7-40) for both child boxes at once (x = left child, y = right child), packed fp32, with the
// per-ray reciprocal hoisted (inv = 1/dir is the same IEEE division the reference redoes per box).  Returns the
// c-independent part: e = max(tmin, t0x, t0y, t0z) and m = min(t1x, t1y, t1z); the box passes iff !(min(c, m) <= e).
// The reference's per-axis early-outs are equivalent to this single final test because its running min only grows
// and its running max only shrinks; NaN t0/t1 are ignored by both forms (comparisons false / fmaxf,fminf return the
// other operand).
__device__ __forceinline__ void box_pair(const BoxPairs &b, V3 o, V3 inv, float &e_l, float &m_l, float &e_r, float &m_r) {
    const f2 t0x = (b.nx - o.x) * inv.x, t1x = (b.fx - o.x) * inv.x;
    const f2 t0y = (b.ny - o.y) * inv.y, t1y = (b.fy - o.y) * inv.y;
    const f2 t0z = (b.nz - o.z) * inv.z, t1z = (b.fz - o.z) * inv.z;
    e_l = fmaxf(fmaxf(fmaxf(0.0f, t0x.x), t0y.x), t0z.x); e_r = fmaxf(fmaxf(fmaxf(0.0f, t0x.y), t0y.y), t0z.y);
    // min(c, t1x, t1y, t1z) = min(c, m) with m = min over the non-NaN t1 (fminf ignores NaNs in any order)
    m_l = fminf(fminf(t1x.x, t1y.x), t1z.x); m_r = fminf(fminf(t1x.y, t1y.y), t1z.y);
}

// bvh.cu:154-160: pop when neither child is to be traversed; otherwise descend left first and push right iff both.
//
// Branch-free (a divergent `if` costs three scalar instructions for the EXEC juggling, and scalar instructions take issue
// slots like vector ones -- profiles/r02/calib_issue_rates.txt): the newest entry lives in the register `top` (-1 when the
// stack is empty), LDS holds entries 0 .. sp-2.  Every step
//   * reads `below` = entry sp-2 (slot sp; the two sentinel slots make that -1 for sp < 2) -- needed only by a pop, but an
//     unconditional read issued before the box arithmetic is off the critical path and cheaper than a branch;
//   * writes `top` to its LDS home, the slot of entry sp-1 -- needed only by a push, harmless otherwise (that slot is dead
//     while the entry lives in the register; with sp = 0 it re-writes the sentinel -1).
// Then  pop : node = top, top = below, sp-1      push : node = lref, top = rref, sp+1      else : node = the one child hit.
// A pop from the empty stack yields node = top = -1 = kTravDone.
// (`sp` is the LDS address of slot `entries in use`; StackRef only tells where an empty stack starts)
template <bool NARROW> constexpr int kStackStride = NARROW ? 128 : 256;      // 64 lanes x 2 or 4 bytes per slot
typedef __attribute__((address_space(3))) char lds_char;
template <bool NARROW>
__device__ __forceinline__ uint32_t stack_base(const StackRef &st) { return NARROW ? (uint32_t)(uintptr_t)st.s16 : (uint32_t)(uintptr_t)st.s32; }
template <bool NARROW>
__device__ __forceinline__ int stack_exchange(uint32_t sp, int top) {
    lds_char *slot = (lds_char *)(uintptr_t)sp;
    int below;
    if (NARROW) { below = (int)*(lds_i16 *)slot; *(lds_i16 *)(slot + 128) = (int16_t)top; }
    else { below = *(lds_i32 *)slot; *(lds_i32 *)(slot + 256) = top; }
    return below;
}
template <bool NARROW>
__device__ __forceinline__ void trav_advance(Trav &tv, bool trav_l, bool trav_r, int lref, int rref, int below) {
    const bool both = trav_l && trav_r, none = !trav_l && !trav_r;
    const int top = tv.top;
    tv.node = trav_l ? lref : (trav_r ? rref : top);
    tv.top = both ? rref : (none ? below : top);
    tv.sp += (uint32_t)(both ? kStackStride<NARROW> : (none ? -kStackStride<NARROW> : 0));
}

// Visit of an INNER record (both children internal): two box tests, no triangle work.  ALL_CACHED: the whole inner tree is
// LDS resident (n_cached == n_inner), no global fall-back path in the step.
template <bool COUNT, bool NARROW, bool ALL_CACHED>
__device__ __forceinline__ void trav_step_inner(Trav &tv, const NodeSrc &ns, V3 o, V3 inv, const StackRef &stack, TravStats &ts) {
    BoxPairs b;
    int lref, rref;
    const int below = stack_exchange<NARROW>(tv.sp, tv.top);
    if (ALL_CACHED) {
        const uint32_t node = (uint32_t)tv.node, rec = node << 4;
        const uint32_t ax = rec + tv.nf[0], ay = rec + tv.nf[1], az = rec + tv.nf[2];
        b.nx = *(lds_cf2 *)(uintptr_t)ax; b.fx = *(lds_cf2 *)(uintptr_t)(ax ^ 8u);
        b.ny = *(lds_cf2 *)(uintptr_t)ay; b.fy = *(lds_cf2 *)(uintptr_t)(ay ^ 8u);
        b.nz = *(lds_cf2 *)(uintptr_t)az; b.fz = *(lds_cf2 *)(uintptr_t)(az ^ 8u);
        const uint32_t r0 = ns.lds_r0[node];
        if (NARROW) { lref = (int)(r0 & 0xffffu); rref = (int)(r0 >> 16); }
        else { lref = (int)r0; rref = (int)ns.lds_r1[node]; }
    } else {
        fetch_node<NARROW>(ns, tv.node, inv, tv.nf, b, lref, rref);
    }
    if (COUNT) { ts.n_iters++; ts.n_box += 2u; }
    float e_l, m_l, e_r, m_r;
    box_pair(b, o, inv, e_l, m_l, e_r, m_r);
    const float c = tv.c;
    const bool trav_l = !(fminf(c, m_l) <= e_l);
    const bool trav_r = !(fminf(c, m_r) <= e_r);
    trav_advance<NARROW>(tv, trav_l, trav_r, lref, rref, below);
}

// A burst of up to eight INNER visits in hand-scheduled gfx950 assembly (production build, 16-bit references, whole inner tree
// in LDS).  Instruction for instruction the visit the compiler makes of trav_step_inner -- the same loads, the same packed
// subtractions / products / min / max / compares in the same operand order, the same selects -- but the bookkeeping around it is
// cheaper than anything the compiler will emit for `if (at_inner) step`:
//   * EXEC only ever shrinks inside a burst (a lane that leaves the inner records does not come back before the next
//     scheduling decision), so it is saved once, narrowed by the v_cmpx that tests `still at an inner record` and restored
//     once -- no s_and_saveexec / s_cbranch_execz / s_or exec per visit;
//   * the two wait states between the compare that writes a lane mask and the select that reads it are filled with work;
//   * LDS returns data in issue order, so three partial s_waitcnt let the x products start while the y and z planes are
//     still in flight (a third of a wave's life was spent in s_waitcnt);
//   * origin and reciprocal direction are read from three register pairs with op_sel broadcasts instead of six splat pairs.
// The burst ends after eight visits or as soon as fewer than `stay` lanes are still at inner records (stay >= 1), exactly like
// the C++ loop in render_kernel.  Registers v100-v118 and s80-s86 are scratch.
#ifndef SRT_ASM_SPLIT_WAIT
#define SRT_ASM_SPLIT_WAIT 1
#endif
// LDS returns data in issue order, so the x products can start while the y / z planes are still in flight
#if SRT_ASM_SPLIT_WAIT == 2
#define SRT_LDS_ORDER_A
#define SRT_LDS_ORDER_B
#define SRT_LDS_ORDER_C "ds_read_i16 v116, %[sp]\n\t" "v_lshl_add_u32 v112, %[node], 2, %[refs]\n\t" "ds_read_b32 v112, v112\n\t"
#define SRT_WAIT_X "s_waitcnt lgkmcnt(6)\n\t"
#define SRT_WAIT_Y "s_waitcnt lgkmcnt(4)\n\t"
#define SRT_WAIT_Z "s_waitcnt lgkmcnt(2)\n\t"
#define SRT_WAIT_END "s_waitcnt lgkmcnt(0)\n\t"
#else
#define SRT_LDS_ORDER_A "ds_read_i16 v116, %[sp]\n\t"
#define SRT_LDS_ORDER_B "v_lshl_add_u32 v112, %[node], 2, %[refs]\n\t" "ds_read_b32 v112, v112\n\t"
#define SRT_LDS_ORDER_C
#define SRT_WAIT_END
#if SRT_ASM_SPLIT_WAIT == 1
#define SRT_WAIT_X "s_waitcnt lgkmcnt(5)\n\t"
#define SRT_WAIT_Y "s_waitcnt lgkmcnt(2)\n\t"
#define SRT_WAIT_Z "s_waitcnt lgkmcnt(0)\n\t"
#else
#define SRT_WAIT_X "s_waitcnt lgkmcnt(0)\n\t"
#define SRT_WAIT_Y
#define SRT_WAIT_Z
#endif
#endif
// Encoding experiments of round 5 (profiles/r05/experiments/select_encodings.txt): eight back-to-back VOP2 selects on one VCC run at 23
// cycles each in a microkernel (srt_calib kinds 28 / 38) against 4.3 for the VOP3 form with VCC as an explicit operand (39) -- but a VOP2
// select between other vector instructions costs 2.5 (kinds 41-43), and in this block the two forms are within noise of each other
// (330.2 vs 331.4 ms on cfg 3, no SDWA: 330.7).  The VOP2 forms stay.
#ifndef SRT_ASM_SEL_E64
#define SRT_ASM_SEL_E64 0
#endif
#if SRT_ASM_SEL_E64
#define SRT_SEL_VCC "v_cndmask_b32_e64"
#else
#define SRT_SEL_VCC "v_cndmask_b32_e32"
#endif
#ifndef SRT_ASM_NO_SDWA
#define SRT_ASM_NO_SDWA 0
#endif
#if SRT_ASM_NO_SDWA
#define SRT_SEL_LREF "v_and_b32 v113, 0xffff, v112\n\t" "v_cndmask_b32_e64 %[node], v118, v113, vcc\n\t"
#else
#define SRT_SEL_LREF "v_cndmask_b32_sdwa %[node], v118, v112, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
#endif
#define SRT_INNER_VISIT_ASM                                                                                              \
    "v_lshlrev_b32 v112, 4, %[node]\n\t"                                                                                 \
    "ds_write_b16 %[sp], %[top] offset:128\n\t"                                                                          \
    "v_add_u32 v113, v112, %[nf0]\n\t"                             /* near pair addresses of the three axes */           \
    "v_add_u32 v114, v112, %[nf1]\n\t"                                                                                   \
    "v_add_u32 v115, v112, %[nf2]\n\t"                                                                                   \
    "ds_read_b64 v[100:101], v113\n\t"                                                                                   \
    "v_xor_b32 v113, 8, v113\n\t"                                  /* far pair = the other half of the plane entry */    \
    SRT_LDS_ORDER_A                                                                                                      \
    "ds_read_b64 v[102:103], v113\n\t"                                                                                   \
    "ds_read_b64 v[104:105], v114\n\t"                                                                                   \
    "v_xor_b32 v114, 8, v114\n\t"                                                                                        \
    SRT_LDS_ORDER_B                                                                                                      \
    "ds_read_b64 v[106:107], v114\n\t"                                                                                   \
    "ds_read_b64 v[108:109], v115\n\t"                                                                                   \
    "v_xor_b32 v115, 8, v115\n\t"                                                                                        \
    "ds_read_b64 v[110:111], v115\n\t"                                                                                   \
    SRT_LDS_ORDER_C                                                                                                      \
    SRT_WAIT_X                                                                                                           \
    "v_pk_add_f32 v[100:101], v[100:101], %[p0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" /* near x - o.x */        \
    "v_pk_add_f32 v[102:103], v[102:103], %[p0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" /* far  x - o.x */        \
    "v_pk_mul_f32 v[100:101], %[p1], v[100:101] op_sel:[1,0]\n\t"                              /* * inv.x */             \
    "v_pk_mul_f32 v[102:103], %[p1], v[102:103] op_sel:[1,0]\n\t"                                                        \
    SRT_WAIT_Y                                                                                                           \
    "v_pk_add_f32 v[106:107], v[106:107], %[p0] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"    /* far  y - o.y */        \
    "v_pk_add_f32 v[104:105], v[104:105], %[p0] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"    /* near y - o.y */        \
    "v_pk_mul_f32 v[106:107], %[p2], v[106:107] op_sel_hi:[0,1]\n\t"                           /* * inv.y */             \
    "v_pk_mul_f32 v[104:105], %[p2], v[104:105] op_sel_hi:[0,1]\n\t"                                                     \
    SRT_WAIT_Z                                                                                                           \
    "v_pk_add_f32 v[108:109], v[108:109], %[p1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" /* near z - o.z */        \
    "v_pk_add_f32 v[110:111], v[110:111], %[p1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" /* far  z - o.z */        \
    "v_pk_mul_f32 v[108:109], %[p2], v[108:109] op_sel:[1,0]\n\t"                              /* * inv.z */             \
    "v_pk_mul_f32 v[110:111], %[p2], v[110:111] op_sel:[1,0]\n\t"                                                        \
    "v_max_f32 v100, 0, v100\n\t"                                                                                        \
    "v_min_f32 v102, v102, v106\n\t"                                                                                     \
    "v_max3_f32 v100, v100, v104, v108\n\t"                        /* e_l */                                             \
    "v_max_f32 v101, 0, v101\n\t"                                                                                        \
    "v_min_f32 v103, v103, v107\n\t"                                                                                     \
    "v_min3_f32 v102, %[c], v102, v110\n\t"                        /* min(c, m_l) */                                     \
    "v_max3_f32 v101, v101, v105, v109\n\t"                        /* e_r */                                             \
    "v_cmp_nle_f32 vcc, v102, v100\n\t"                            /* trav_l */                                          \
    "v_min3_f32 v100, %[c], v103, v111\n\t"                        /* min(c, m_r) */                                     \
    "v_cmp_nle_f32_e64 s[82:83], v100, v101\n\t"                   /* trav_r */                                          \
    SRT_WAIT_END                                                                                                         \
    "v_lshrrev_b32 v117, 16, v112\n\t"                             /* rref */                                            \
    "s_or_b64 s[84:85], vcc, s[82:83]\n\t"                         /* any */                                             \
    "v_cndmask_b32_e64 v118, %[top], v117, s[82:83]\n\t"                                                                 \
    SRT_SEL_LREF                                                                                                         \
    "s_and_b64 vcc, vcc, s[82:83]\n\t"                             /* both */                                            \
    "v_cndmask_b32_e64 v118, %[ms], 0, s[84:85]\n\t"                                                                     \
    "v_cndmask_b32_e64 %[top], v116, %[top], s[84:85]\n\t"                                                               \
    SRT_SEL_VCC " v118, v118, %[ps], vcc\n\t"                                                                            \
    SRT_SEL_VCC " %[top], %[top], v117, vcc\n\t"                                                                         \
    "v_add_u32 %[sp], v118, %[sp]\n\t"
#ifndef SRT_ASM_CMPX
#define SRT_ASM_CMPX 1
#endif
#if SRT_ASM_CMPX
// v_cmpx narrows EXEC itself (and writes VCC, which nothing reads)
#define SRT_INNER_NEXT_ASM                                                                                               \
    "v_cmpx_gt_u32 vcc, %[ninner], %[node]\n\t"                                                                          \
    "s_bcnt1_i32_b64 s86, exec\n\t"                                                                                      \
    "s_cmp_lt_u32 s86, %[stay]\n\t"                                                                                      \
    "s_cbranch_scc1 .Lsrt_burst_end%=\n\t"
#else
#define SRT_INNER_NEXT_ASM                                                                                               \
    "v_cmp_gt_u32 vcc, %[ninner], %[node]\n\t"                                                                           \
    "s_bcnt1_i32_b64 s86, vcc\n\t"                                                                                       \
    "s_cmp_lt_u32 s86, %[stay]\n\t"                                                                                      \
    "s_cbranch_scc1 .Lsrt_burst_end%=\n\t"                                                                               \
    "s_mov_b64 exec, vcc\n\t"
#endif
__device__ __forceinline__ void inner_burst8_asm(Trav &tv, const NodeSrc &ns, V3 o, V3 inv, uint32_t n_inner, uint32_t stay) {
    const f2 p0 = mk2(o.x, o.y), p1 = mk2(o.z, inv.x), p2 = mk2(inv.y, inv.z);
    const int minus_stride = -kStackStride<true>, plus_stride = kStackStride<true>;
    const uint32_t refs = (uint32_t)(uintptr_t)ns.lds_r0;
    asm volatile(
        "s_mov_b64 s[80:81], exec\n\t"
        "v_cmpx_gt_u32 vcc, %[ninner], %[node]\n\t"
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT_ASM
        SRT_INNER_VISIT_ASM
        ".Lsrt_burst_end%=:\n\t"
        "s_mov_b64 exec, s[80:81]\n\t"
        : [node] "+v"(tv.node), [top] "+v"(tv.top), [sp] "+v"(tv.sp)
        : [nf0] "v"(tv.nf[0]), [nf1] "v"(tv.nf[1]), [nf2] "v"(tv.nf[2]), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [c] "v"(tv.c),
          [ms] "v"(minus_stride), [ps] "v"(plus_stride), [ninner] "s"(n_inner), [refs] "s"(refs), [stay] "s"(stay)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114",
          "v115", "v116", "v117", "v118", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "vcc", "scc", "memory");
}

// The scheduling decision of render_kernel's traversal phase together with the INNER bursts it leads to, in one block:
//   repeat { count the lanes waiting for a shading pass / at a fringe record / at an inner record; weigh them (the same
//            unsigned arithmetic as the C++ loop: waiting * score_shade, fringe * score_fringe, inner << 8);
//            shading wins -> return 0;  fringe wins -> return 1;  else one burst of up to eight INNER visits }
// so that a burst costs one decision (14 scalar instructions) and no trip through compiler-generated control flow; the
// caller only runs the FRINGE visit (C++) when 1 comes back.  `stay` of a burst = ceil(inner lanes / 3) (kBurstDrop = 3).
#define SRT_INNER_NEXT2_ASM                                                                                              \
    "v_cmpx_gt_u32 vcc, %[ninner], %[node]\n\t"                                                                          \
    "s_bcnt1_i32_b64 s86, exec\n\t"                                                                                      \
    "s_cmp_lt_u32 s86, s87\n\t"                                                                                          \
    "s_cbranch_scc1 .Lsrt_phase_burst_end%=\n\t"
__device__ __forceinline__ uint32_t inner_phase_asm(Trav &tv, const NodeSrc &ns, V3 o, V3 inv, uint32_t n_inner, uint32_t n_alive,
                                                    uint32_t score_shade, uint32_t score_fringe) {
    const f2 p0 = mk2(o.x, o.y), p1 = mk2(o.z, inv.x), p2 = mk2(inv.y, inv.z);
    const int minus_stride = -kStackStride<true>, plus_stride = kStackStride<true>;
    const uint32_t refs = (uint32_t)(uintptr_t)ns.lds_r0;
    uint32_t kind;
    asm volatile(
        "s_mov_b64 s[80:81], exec\n\t"
        ".Lsrt_phase_decide%=:\n\t"
        "v_cmp_lt_i32 vcc, -1, %[node]\n\t"                          /* traversing */
        "v_cmp_le_i32_e64 s[82:83], %[ninner], %[node]\n\t"          /* ... at a fringe record */
        "s_bcnt1_i32_b64 s86, vcc\n\t"
        "s_bcnt1_i32_b64 s87, s[82:83]\n\t"
        "s_sub_u32 s88, %[nalive], s86\n\t"                          /* lanes waiting for a shading pass */
        "s_sub_u32 s86, s86, s87\n\t"                                /* lanes at an inner record */
        "s_mul_i32 s88, s88, %[wshade]\n\t"
        "s_mul_i32 s89, s87, %[wfringe]\n\t"
        "s_lshl_b32 s84, s86, 8\n\t"
        "s_max_u32 s85, s89, s84\n\t"
        "s_mov_b32 %[kind], 0\n\t"
        "s_cmp_gt_u32 s88, s85\n\t"
        "s_cbranch_scc1 .Lsrt_phase_end%=\n\t"                       /* shading pass (or nothing traversing) */
        "s_mov_b32 %[kind], 1\n\t"
        "s_cmp_gt_u32 s89, s84\n\t"
        "s_cbranch_scc1 .Lsrt_phase_end%=\n\t"                       /* FRINGE visit */
        "s_add_u32 s87, s86, 2\n\t"                                  /* stay = ceil(inner lanes / 3) */
        "s_mul_hi_u32 s87, s87, 0xaaaaaaab\n\t"
        "s_lshr_b32 s87, s87, 1\n\t"
        "s_andn2_b64 exec, vcc, s[82:83]\n\t"                        /* the lanes at inner records */
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT2_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT2_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT2_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT2_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT2_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT2_ASM
        SRT_INNER_VISIT_ASM SRT_INNER_NEXT2_ASM
        SRT_INNER_VISIT_ASM
        ".Lsrt_phase_burst_end%=:\n\t"
        "s_mov_b64 exec, s[80:81]\n\t"
        "s_branch .Lsrt_phase_decide%=\n\t"
        ".Lsrt_phase_end%=:\n\t"
        : [node] "+v"(tv.node), [top] "+v"(tv.top), [sp] "+v"(tv.sp), [kind] "=&s"(kind)
        : [nf0] "v"(tv.nf[0]), [nf1] "v"(tv.nf[1]), [nf2] "v"(tv.nf[2]), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [c] "v"(tv.c),
          [ms] "v"(minus_stride), [ps] "v"(plus_stride), [ninner] "s"(n_inner), [refs] "s"(refs), [nalive] "s"(n_alive),
          [wshade] "s"(score_shade), [wfringe] "s"(score_fringe)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114",
          "v115", "v116", "v117", "v118", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "vcc",
          "scc", "memory");
    return kind;
}

// A burst of up to four INNER visits for a tree that is only PARTLY LDS resident (32-bit references; production build of
// render_kernel<0, false, false, .>), hand-scheduled like inner_burst8_asm.  The compiler's form of this visit -- `if (cached) LDS else
// memory` around one shared arithmetic block -- is ~100 instructions: three register copies per visit for the loop-carried state, the
// memory path's twelve near / far selects, a full `s_waitcnt vmcnt(0)` BEFORE the first LDS read of the other lanes, an exec juggle per
// region.  Here (~72):
//   * the lanes beyond the LDS prefix load a PRE-SWIZZLED copy of their record (80 B: per axis lo_L lo_R hi_L hi_R lo_L lo_R; a
//     16-byte load at +0 or +8 IS (near pair, far pair)) -- the selection of aabb.cu:21-25 by address, like the LDS lanes' -- so both
//     kinds of lanes fill the same twelve registers and the arithmetic block is shared without a single select;
//   * the memory loads are issued first, the LDS reads of the other lanes go out while they fly, one wait pair before the arithmetic;
//   * EXEC is saved once per burst and narrowed by v_cmpx like in inner_burst8_asm; the two sub-masks of a visit cost one s_and_saveexec
//     and one s_andn2.
// Arithmetic: instruction for instruction trav_step_inner's (same operands, same order).  Registers v100-v118, s80-s91 are scratch.
#define SRT_INNER_VISIT_MIXED_ASM                                                                                        \
    "ds_write_b32 %[sp], %[top] offset:256\n\t"                                                                          \
    "ds_read_b32 v116, %[sp]\n\t"                                   /* below */                                           \
    "v_cmp_le_u32 vcc, %[ncached], %[node]\n\t"                     /* lanes beyond the LDS prefix */                     \
    "s_and_saveexec_b64 s[90:91], vcc\n\t"                                                                               \
    "s_cbranch_execz 1f\n\t"                                                                                             \
    "v_mul_u32_u24 v117, 0x50, %[node]\n\t"                         /* 80-byte records */                                 \
    "v_add_u32 v114, v117, %[g0]\n\t"                                                                                    \
    "v_add_u32 v115, v117, %[g1]\n\t"                                                                                    \
    "v_add_u32 v118, v117, %[g2]\n\t"                                                                                    \
    "buffer_load_dwordx4 v[100:103], v114, %[rsrc], 0 offen\n\t"                                                         \
    "buffer_load_dwordx4 v[104:107], v115, %[rsrc], 0 offen offset:24\n\t"                                               \
    "buffer_load_dwordx4 v[108:111], v118, %[rsrc], 0 offen offset:48\n\t"                                               \
    "buffer_load_dwordx2 v[112:113], v117, %[rsrc], 0 offen offset:72\n\t"                                               \
    "1:\n\t"                                                                                                             \
    "s_andn2_b64 exec, s[90:91], vcc\n\t"                           /* lanes inside the prefix */                         \
    "s_cbranch_execz 2f\n\t"                                                                                             \
    "v_lshlrev_b32 v117, 4, %[node]\n\t"                                                                                 \
    "v_add_u32 v114, v117, %[nf0]\n\t"                                                                                   \
    "v_add_u32 v115, v117, %[nf1]\n\t"                                                                                   \
    "v_add_u32 v118, v117, %[nf2]\n\t"                                                                                   \
    "ds_read_b64 v[100:101], v114\n\t"                                                                                   \
    "v_xor_b32 v114, 8, v114\n\t"                                                                                        \
    "ds_read_b64 v[102:103], v114\n\t"                                                                                   \
    "ds_read_b64 v[104:105], v115\n\t"                                                                                   \
    "v_xor_b32 v115, 8, v115\n\t"                                                                                        \
    "ds_read_b64 v[106:107], v115\n\t"                                                                                   \
    "ds_read_b64 v[108:109], v118\n\t"                                                                                   \
    "v_xor_b32 v118, 8, v118\n\t"                                                                                        \
    "ds_read_b64 v[110:111], v118\n\t"                                                                                   \
    "v_lshl_add_u32 v117, %[node], 2, %[refs0]\n\t"                                                                      \
    "ds_read_b32 v112, v117\n\t"                                    /* lref */                                            \
    "v_lshl_add_u32 v117, %[node], 2, %[refs1]\n\t"                                                                      \
    "ds_read_b32 v113, v117\n\t"                                    /* rref */                                            \
    "2:\n\t"                                                                                                             \
    "s_mov_b64 exec, s[90:91]\n\t"                                                                                       \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                                  \
    "v_pk_add_f32 v[100:101], v[100:101], %[p0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" /* near x - o.x */        \
    "v_pk_add_f32 v[102:103], v[102:103], %[p0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" /* far  x - o.x */        \
    "v_pk_mul_f32 v[100:101], %[p1], v[100:101] op_sel:[1,0]\n\t"                              /* * inv.x */             \
    "v_pk_mul_f32 v[102:103], %[p1], v[102:103] op_sel:[1,0]\n\t"                                                        \
    "v_pk_add_f32 v[106:107], v[106:107], %[p0] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"    /* far  y - o.y */        \
    "v_pk_add_f32 v[104:105], v[104:105], %[p0] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"    /* near y - o.y */        \
    "v_pk_mul_f32 v[106:107], %[p2], v[106:107] op_sel_hi:[0,1]\n\t"                           /* * inv.y */             \
    "v_pk_mul_f32 v[104:105], %[p2], v[104:105] op_sel_hi:[0,1]\n\t"                                                     \
    "v_pk_add_f32 v[108:109], v[108:109], %[p1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" /* near z - o.z */        \
    "v_pk_add_f32 v[110:111], v[110:111], %[p1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" /* far  z - o.z */        \
    "v_pk_mul_f32 v[108:109], %[p2], v[108:109] op_sel:[1,0]\n\t"                              /* * inv.z */             \
    "v_pk_mul_f32 v[110:111], %[p2], v[110:111] op_sel:[1,0]\n\t"                                                        \
    "v_max_f32 v100, 0, v100\n\t"                                                                                        \
    "v_min_f32 v102, v102, v106\n\t"                                                                                     \
    "v_max3_f32 v100, v100, v104, v108\n\t"                        /* e_l */                                             \
    "v_max_f32 v101, 0, v101\n\t"                                                                                        \
    "v_min_f32 v103, v103, v107\n\t"                                                                                     \
    "v_min3_f32 v102, %[c], v102, v110\n\t"                        /* min(c, m_l) */                                     \
    "v_max3_f32 v101, v101, v105, v109\n\t"                        /* e_r */                                             \
    "v_cmp_nle_f32 vcc, v102, v100\n\t"                            /* trav_l */                                          \
    "v_min3_f32 v100, %[c], v103, v111\n\t"                        /* min(c, m_r) */                                     \
    "v_cmp_nle_f32_e64 s[82:83], v100, v101\n\t"                   /* trav_r */                                          \
    "s_or_b64 s[84:85], vcc, s[82:83]\n\t"                         /* any */                                             \
    "v_cndmask_b32_e64 v118, %[top], v113, s[82:83]\n\t"                                                                 \
    "v_cndmask_b32_e32 %[node], v118, v112, vcc\n\t"                                                                     \
    "s_and_b64 vcc, vcc, s[82:83]\n\t"                             /* both */                                            \
    "v_cndmask_b32_e64 v118, %[ms], 0, s[84:85]\n\t"                                                                     \
    "v_cndmask_b32_e64 %[top], v116, %[top], s[84:85]\n\t"                                                               \
    SRT_SEL_VCC " v118, v118, %[ps], vcc\n\t"                                                                            \
    SRT_SEL_VCC " %[top], %[top], v113, vcc\n\t"                                                                         \
    "v_add_u32 %[sp], v118, %[sp]\n\t"
#define SRT_INNER_NEXT_MIXED_ASM                                                                                         \
    "v_cmpx_gt_u32 vcc, %[ninner], %[node]\n\t"                                                                          \
    "s_bcnt1_i32_b64 s86, exec\n\t"                                                                                      \
    "s_cmp_lt_u32 s86, %[stay]\n\t"                                                                                      \
    "s_cbranch_scc1 .Lsrt_mixed_end%=\n\t"
__device__ __forceinline__ void inner_burst4_mixed_asm(Trav &tv, const NodeSrc &ns, V3 o, V3 inv, uint32_t n_inner, uint32_t stay) {
    const f2 p0 = mk2(o.x, o.y), p1 = mk2(o.z, inv.x), p2 = mk2(inv.y, inv.z);
    const int minus_stride = -kStackStride<false>, plus_stride = kStackStride<false>;
    const uint32_t refs0 = (uint32_t)(uintptr_t)ns.lds_r0, refs1 = (uint32_t)(uintptr_t)ns.lds_r1;
    // byte offset of the (near, far) image inside an axis entry of the pre-swizzled record: 0 for a positive direction, 8 for a negative
    // one -- bit 3 of the ray's LDS plane addresses says which (the planes are 16-byte aligned)
    const uint32_t g0 = tv.nf[0] & 8u, g1 = tv.nf[1] & 8u, g2 = tv.nf[2] & 8u;
    asm volatile(
        "s_mov_b64 s[80:81], exec\n\t"
        "v_cmpx_gt_u32 vcc, %[ninner], %[node]\n\t"
        SRT_INNER_VISIT_MIXED_ASM SRT_INNER_NEXT_MIXED_ASM
        SRT_INNER_VISIT_MIXED_ASM SRT_INNER_NEXT_MIXED_ASM
        SRT_INNER_VISIT_MIXED_ASM SRT_INNER_NEXT_MIXED_ASM
        SRT_INNER_VISIT_MIXED_ASM
        ".Lsrt_mixed_end%=:\n\t"
        "s_mov_b64 exec, s[80:81]\n\t"
        : [node] "+v"(tv.node), [top] "+v"(tv.top), [sp] "+v"(tv.sp)
        : [nf0] "v"(tv.nf[0]), [nf1] "v"(tv.nf[1]), [nf2] "v"(tv.nf[2]), [g0] "v"(g0), [g1] "v"(g1), [g2] "v"(g2), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2),
          [c] "v"(tv.c), [ms] "v"(minus_stride), [ps] "v"(plus_stride), [ninner] "s"(n_inner), [ncached] "s"((uint32_t)ns.n_cached), [refs0] "s"(refs0),
          [refs1] "s"(refs1), [stay] "s"(stay), [rsrc] "s"(ns.global_nodes_sw)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114",
          "v115", "v116", "v117", "v118", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s90", "s91", "vcc", "scc", "memory");
}

// Visit of a FRINGE record (at least one leaf child), in two halves so that a caller can put independent work (a burst of
// INNER steps for other lanes of the wave) between the loads and their first use: the record comes from L2, several hundred
// cycles away.
struct FringeFetch { f4v q0, q1, q2, q3, q4, q5; int below; };
template <bool NARROW>
__device__ __forceinline__ void trav_fringe_fetch(FringeFetch &ff, const Trav &tv, const NodeSrc &ns, const StackRef &stack) {
    // one round of independent loads: 12 (left, right) pairs = six 16-byte loads (record layout: flatten_scene)
    const uint32_t off = __umul24((uint32_t)(tv.node - ns.n_inner), ns.fringe_stride);   // (full-rate 24-bit multiply; < 2^24 fringe records)
    ff.q0 = buf_load16(ns.global_fringe, off); ff.q1 = buf_load16(ns.global_fringe, off + 16u); ff.q2 = buf_load16(ns.global_fringe, off + 32u);
    ff.q3 = buf_load16(ns.global_fringe, off + 48u); ff.q4 = buf_load16(ns.global_fringe, off + 64u); ff.q5 = buf_load16(ns.global_fringe, off + 80u);
    // a FRINGE visit never pushes (at most one child is internal), so it only needs the entry a pop would bring up
    ff.below = NARROW ? (int)*(lds_i16 *)(uintptr_t)tv.sp : *(lds_i32 *)(uintptr_t)tv.sp;
}
// PAIRED: the tree has no node with exactly ONE leaf child (srt_scene_is_paired: the SAH builder's even splits), so every FRINGE record
// holds two triangles: the slab test of "the internal child" and the descent into it -- a fifth of the visit's instructions, needed by
// 5-9 % of the FRINGE visits of an unpaired tree -- are not compiled in, and the visit always ends with a pop.
template <bool COUNT, bool NARROW, bool PAIRED = false>
__device__ __forceinline__ void trav_fringe_compute(const FringeFetch &ff, Trav &tv, V3 o, V3 d, V3 inv, TravStats &ts) {
    const f4v q0 = ff.q0, q1 = ff.q1, q2 = ff.q2, q3 = ff.q3, q4 = ff.q4, q5 = ff.q5;
    const int below = ff.below;
    const f2 w0 = mk2(q0.x, q0.y), w1 = mk2(q0.z, q0.w), w2 = mk2(q1.x, q1.y), w3 = mk2(q1.z, q1.w), w4 = mk2(q2.x, q2.y), w5 = mk2(q2.z, q2.w);
    const f2 w6 = mk2(q3.x, q3.y), w7 = mk2(q3.z, q3.w), w8 = mk2(q4.x, q4.y), w9 = mk2(q4.z, q4.w);
    const uint32_t fl = __float_as_uint(q5.x), fr = __float_as_uint(q5.y);
    const int lref = (int)__float_as_uint(q5.z), rref = (int)__float_as_uint(q5.w);
    const bool leaf_l = PAIRED || lref < 0, leaf_r = PAIRED || rref < 0;
    if (COUNT) { ts.n_iters++; ts.n_tri += (leaf_l ? 1u : 0u) + (leaf_r ? 1u : 0u); ts.n_box += (leaf_l ? 0u : 1u) + (leaf_r ? 0u : 1u); }

    // ---- the box of the internal child (words 0-5 = xmin xmax ymin ymax zmin zmax of its block).  A FRINGE record has at
    // least one leaf child, so at most ONE child has a box: one scalar slab test on the right child's words when the right
    // child is internal, on the left child's otherwise (both leaves: the result is ignored) -- half the arithmetic of the
    // paired test of an INNER record.  aabb::hit (aabb.cu:7-40) as in box_pair: e = max(tmin, t0x, t0y, t0z), m = min(t1x, t1y, t1z).
    float e_box = 0.f, m_box = 0.f;
    if (!PAIRED) {
        const bool int_r = !leaf_r;
        const float xlo = int_r ? w0.y : w0.x, xhi = int_r ? w1.y : w1.x, ylo = int_r ? w2.y : w2.x, yhi = int_r ? w3.y : w3.x;
        const float zlo = int_r ? w4.y : w4.x, zhi = int_r ? w5.y : w5.x;
        const bool px = inv.x >= 0, py = inv.y >= 0, pz = inv.z >= 0;
        const float t0x = ((px ? xlo : xhi) - o.x) * inv.x, t1x = ((px ? xhi : xlo) - o.x) * inv.x;
        const float t0y = ((py ? ylo : yhi) - o.y) * inv.y, t1y = ((py ? yhi : ylo) - o.y) * inv.y;
        const float t0z = ((pz ? zlo : zhi) - o.z) * inv.z, t1z = ((pz ? zhi : zlo) - o.z) * inv.z;
        e_box = fmaxf(fmaxf(fmaxf(0.0f, t0x), t0y), t0z);
        m_box = fminf(fminf(t1x, t1y), t1z);
    }

    // ---- both leaf triangles in one segment (words 0-9 = n.x n.y n.z D v0w v0h v1w v1h v2w v2h; a fringe record has at
    // least one leaf, the other side's values are ignored) ----------------------------------------------------------------
    const f2 denom = w0 * d.x + w1 * d.y + w2 * d.z;                      // dot(normal, dir), tri.cu:9
    const f2 num = w3 - (w0 * o.x + w1 * o.y + w2 * o.z);                 // D - dot(normal, origin), tri.cu:17
    const f2 t = mk2(num.x / denom.x, num.y / denom.y);
    const bool wyl = fl & kTriWIsY, hzl = fl & kTriHIsZ, wyr = fr & kTriWIsY, hzr = fr & kTriHIsZ;
    // intersection = orig + t*dir (ray.cuh:31-34), the two projected components
    const f2 pw = mk2(wyl ? o.y : o.x, wyr ? o.y : o.x) + t * mk2(wyl ? d.y : d.x, wyr ? d.y : d.x);
    const f2 ph = mk2(hzl ? o.z : o.y, hzr ? o.z : o.y) + t * mk2(hzl ? d.z : d.y, hzr ? d.z : d.y);
    // double_signed_area_2D(v1,v2,v3) = (v1[w]-v3[w])*(v2[h]-v3[h]) - (v2[w]-v3[w])*(v1[h]-v3[h])   (tri.cu:181)
    const f2 a1 = (pw - w6) * (w5 - w7) - (w4 - w6) * (ph - w7);     // (p, v0, v1)
    const f2 a2 = (pw - w8) * (w7 - w9) - (w6 - w8) * (ph - w9);     // (p, v1, v2)
    const f2 a3 = (pw - w4) * (w9 - w5) - (w8 - w4) * (ph - w5);     // (p, v2, v0)
    // is_interior_faster (tri.cu:121-128): all three areas >= 0 for a clockwise triangle, all <= 0 otherwise.  `a <= 0` is
    // `-a >= 0` for every float (zeros of either sign pass both, NaN fails both), and -a is a flip of the sign bit, which the
    // record holds in bit 31 of the flags word for counter-clockwise triangles: one code path, no branch.
    const uint32_t sl = fl & 0x80000000u, sr = fr & 0x80000000u;
    // (all three >= 0  <=>  their NaN-propagating minimum >= 0)
    // (two steps, pinned: holding all three flipped areas of both triangles for one three-operand minimum costs registers the
    // kernel does not have)
    float m_l = inside_min2(a1.x, a2.x, sl);
    float m_r = inside_min2(a1.y, a2.y, sr);
    asm volatile("" : "+v"(m_l), "+v"(m_r));
    const bool in_l = inside_min3(m_l, a3.x, sl);
    const bool in_r = inside_min3(m_r, a3.y, sr);
    // plane not parallel, t >= tmin, inside (everything but `t <= c`)
    const bool ok_l = leaf_l & !(fabsf(denom.x) < 1e-8f) & (0.0f <= t.x) & in_l;
    const bool ok_r = leaf_r & !(fabsf(denom.y) < 1e-8f) & (0.0f <= t.y) & in_r;

    // ---- decisions in the reference's order (bvh.cu:128-160): left child with c, right child with the updated c -----------
    const float c0 = tv.c;
    const bool hit_l = ok_l & (t.x <= c0);
    const float c1 = hit_l ? t.x : c0;
    const bool hit_r = ok_r & (t.y <= c1);
    const float c2 = hit_r ? t.y : c1;
    // the one box: an internal LEFT child is tested with c0 (before anything on the right), an internal RIGHT child with c1 (after
    // the left leaf's hit, if any)
    tv.c = c2;
    tv.hit = hit_r ? ~rref : (hit_l ? ~lref : tv.hit);
    const int top = tv.top;
    if (PAIRED) {      // two leaves: nothing to descend into (bvh.cu:154-160: neither child is traversable -> pop)
        tv.node = top;
        tv.top = below;
        tv.sp += (uint32_t)(-kStackStride<NARROW>);
        return;
    }
    const bool pass = !(fminf(leaf_r ? c0 : c1, m_box) <= e_box);
    // bvh.cu:154-160 with at most one traversable child: descend into it, or pop; never a push
    const bool go = !(leaf_l & leaf_r) & pass;
    tv.node = go ? (leaf_r ? lref : rref) : top;
    tv.top = go ? top : below;
    tv.sp += (uint32_t)(go ? 0 : -kStackStride<NARROW>);
}
template <bool COUNT, bool NARROW, bool PAIRED = false>
__device__ __forceinline__ void trav_step_fringe(Trav &tv, const NodeSrc &ns, V3 o, V3 d, V3 inv, const StackRef &stack, TravStats &ts) {
    FringeFetch ff;
    trav_fringe_fetch<NARROW>(ff, tv, ns, stack);
    trav_fringe_compute<COUNT, NARROW, PAIRED>(ff, tv, o, d, inv, ts);
}


}  // namespace srt
