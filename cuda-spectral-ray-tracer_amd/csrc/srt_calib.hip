// srt_calib.hip -- issue-rate calibration microkernels for gfx950 (the denominator of bench.py's roofline).
//
// The render kernel is a divergent VALU/LDS program, not a memory stream: what bounds it is how many wave64 vector
// instructions a SIMD can issue per cycle with the four resident waves this kernel runs (127 VGPRs -> 4 waves / SIMD).
// That rate is measured here instead of assumed: each kernel runs one long loop of independent (or deliberately
// dependent) instructions written in inline asm, every wave stamps s_memtime around its loop, and the host adds a
// HIP-event wall time so that cycles turn into seconds at the clock the chip really holds under that load.
//
// kinds (8 instructions per asm block, 4 blocks per loop trip):
//   0 v_add_f32, 8 independent accumulators          1 v_pk_mul_f32, 8 independent accumulator pairs
//   2 v_fma_f32, 8 independent accumulators          3 v_add_f32, ONE dependent chain
//   4 s_add_u32, 8 independent scalars               5 4 x (v_add_f32 + s_add_u32) interleaved (co-issue)
//   6 v_cmp_lt_f32 + v_cndmask_b32 pairs (VCC)       7 ds_read_b64, lane-linear addresses (conflict free)
//   8 ds_read_b64, per-lane random 16-byte records   9 v_max3_f32, 8 independent accumulators
//  10 v_add_f32 with only 26 of 64 lanes enabled (EXEC-masked: does a partially filled instruction cost less?)
//  11 4 x buffer_load_dwordx4 of a random 64-byte record per lane (16 MB table: L2 / MALL served), all 64 lanes
//  12 the same with 16 lanes enabled, CONTIGUOUS (lanes 0-15: four full quads)
//  13 the same with 16 lanes enabled, ONE PER QUAD (every fourth lane)   -- does the texture-address unit skip empty quads?
//  14 the same with 32 contiguous lanes                                   15 the same with 32 lanes, two per quad
//  16 / 17 / 18  4 x buffer_load_dword / dwordx2 / dwordx3 at the same four offsets of the record, all 64 lanes -- does the cost of a
//                vector-memory instruction in the CU's memory front end scale with its width?          19 / 20 / 21 the same, 16 lanes
//  22 three dwordx4 + one dword (a 52-byte record), 16 lanes
//  23-35 (round 5) what makes a vector instruction expensive -- its ENCODING (8 bytes: VOP3, a 32-bit literal, SDWA) or its operands?
//  23 v_add_f32_e64 (two sources, VOP3 encoding)        24 v_add_f32 with a 32-bit literal (VOP2 + literal = 8 bytes)
//  25 v_fmac_f32 (VOP2: 4 bytes, three register reads)   26 v_mov_b32                 27 v_cndmask_b32_e64 (mask in an SGPR pair)
//  28 v_cndmask_b32_e32 (mask in VCC)                    29 v_mul_f32 with an SGPR source (4 bytes)   30 v_xor_b32
//  31 v_pk_add_f32                                       32 v_rcp_f32                 33 v_cvt_f32_u32
//  34 v_add_f32 SDWA (8 bytes, two sources)              35 v_fmamk_f32 (VOP2 + literal, three reads)
//  36 v_cmp_lt_f32 -> VCC     37 v_cmp_lt_f32_e64 -> SGPR pair     38 s_mov vcc + 8 x v_cndmask_b32_e32     39 v_cndmask_b32_e64 with VCC     40 v_add_u32 with an SGPR source
//  41-46 VOP2 selects (implicit VCC) in context: alternating with v_add; v_cmp + two consumers; v_cmp, three adds, consumer; s_and vcc + two VOP2
//        consumers + six adds (the tail of the INNER visit); the same with VOP3-encoded selects; one stale-VCC select in eight
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "srt_internal.h"

namespace srt {

constexpr int kCalibKinds = 47;

#define REP4(x) x x x x

template <int KIND>
__global__ __launch_bounds__(1024) void calib_kernel(uint32_t iters, float *sink, unsigned long long *cycles, const float4 *table, uint32_t n_records) {
    extern __shared__ float4 lds4[];
    const uint32_t tid = threadIdx.x;
    float a0 = (float)tid, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
    const float k = 1.0000001f;
    const f2 k2 = {k, k};
    uint32_t s0 = iters, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    // LDS image for kinds 7 / 8: 64 KB of floats
    if (KIND == 7 || KIND == 8) {
        float *l = reinterpret_cast<float *>(lds4);
        for (uint32_t i = tid; i < 16384u; i += blockDim.x) l[i] = (float)i;
        __syncthreads();
    }
    uint32_t lin_addr = (tid & 63u) * 8u;                               // conflict-free ds_read_b64
    uint32_t rnd_addr = ((tid * 2654435761u) >> 20) * 16u;              // 4096 records of 16 B, hashed per lane
    unsigned long long saved_exec = 0;
    float sk = 1.0000001f; unsigned long long smask = 0x5555aaaa3333ccccull;
    asm volatile("" : "+s"(sk), "+s"(smask));
    if (KIND == 10) asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, 0x3ffffff" : "=s"(saved_exec));
    // kinds 11-15: random 64-byte records through a buffer descriptor, like the INNER visit of a tree that does not fit LDS
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(table), 0, (int)(n_records * 64u), 0x00020000);
    uint32_t rnd = (blockIdx.x * blockDim.x + tid) * 2654435761u + 12345u;
    u4v g0 = {0, 0, 0, 0}, g1 = g0, g2 = g0, g3 = g0;
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    typedef unsigned int u3v __attribute__((ext_vector_type(3)));
    unsigned int h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    u2v i0 = {0, 0}, i1 = i0, i2 = i0, i3 = i0;
    u3v j0 = {0, 0, 0}, j1 = j0, j2 = j0, j3 = j0;
    if (KIND == 12 || (KIND >= 19 && KIND <= 22)) asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, 0xffff" : "=s"(saved_exec));
    if (KIND == 13) asm volatile("s_mov_b64 %0, exec\n s_mov_b32 exec_lo, 0x11111111\n s_mov_b32 exec_hi, 0x11111111" : "=s"(saved_exec));
    if (KIND == 14) asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, 0xffffffff" : "=s"(saved_exec));
    if (KIND == 15) asm volatile("s_mov_b64 %0, exec\n s_mov_b32 exec_lo, 0x33333333\n s_mov_b32 exec_hi, 0x33333333" : "=s"(saved_exec));
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (uint32_t i = 0; i < iters; i++) {
        if (KIND == 0 || KIND == 10) {
            REP4(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if (KIND == 1) {
            REP4(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(k2));)
        } else if (KIND == 2) {
            REP4(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        } else if (KIND == 3) {
            REP4(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                              "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                              : "+v"(a0) : "v"(k));)
        } else if (KIND == 4) {
            REP4(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                              "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
                              : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)::"scc");)
        } else if (KIND == 5) {
            REP4(asm volatile("v_add_f32 %0, %0, %8\n s_add_u32 %4, %4, 1\n v_add_f32 %1, %1, %8\n s_add_u32 %5, %5, 1\n"
                              "v_add_f32 %2, %2, %8\n s_add_u32 %6, %6, 1\n v_add_f32 %3, %3, %8\n s_add_u32 %7, %7, 1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "v"(k) : "scc");)
        } else if (KIND == 6) {
            REP4(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n"
                              "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "vcc");)
        } else if (KIND == 7 || KIND == 8) {
            const uint32_t addr = KIND == 7 ? lin_addr : rnd_addr;
            REP4(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                              "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
                              "s_waitcnt lgkmcnt(0)\n"
                              : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7) : "v"(addr) : "memory");)
        } else if (KIND >= 11 && KIND <= 15) {
            // 8 rounds of 4 loads = 32 vector-memory instructions per trip; two rounds in flight
#pragma unroll
            for (int rr = 0; rr < 8; rr++) {
                rnd = rnd * 1664525u + 1013904223u;
                const uint32_t off = ((rnd >> 7) % n_records) * 64u;
                asm volatile("buffer_load_dwordx4 %0, %4, %5, 0 offen\n buffer_load_dwordx4 %1, %4, %5, 0 offen offset:16\n"
                             "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:32\n buffer_load_dwordx4 %3, %4, %5, 0 offen offset:48\n"
                             "s_waitcnt vmcnt(4)\n"
                             : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3) : "v"(off), "s"(rsrc) : "memory");
            }
        } else if (KIND == 16 || KIND == 19) {
#pragma unroll
            for (int rr = 0; rr < 8; rr++) {
                rnd = rnd * 1664525u + 1013904223u;
                const uint32_t off = ((rnd >> 7) % n_records) * 64u;
                asm volatile("buffer_load_dword %0, %4, %5, 0 offen\n buffer_load_dword %1, %4, %5, 0 offen offset:16\n"
                             "buffer_load_dword %2, %4, %5, 0 offen offset:32\n buffer_load_dword %3, %4, %5, 0 offen offset:48\n"
                             "s_waitcnt vmcnt(4)\n"
                             : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3) : "v"(off), "s"(rsrc) : "memory");
            }
        } else if (KIND == 17 || KIND == 20) {
#pragma unroll
            for (int rr = 0; rr < 8; rr++) {
                rnd = rnd * 1664525u + 1013904223u;
                const uint32_t off = ((rnd >> 7) % n_records) * 64u;
                asm volatile("buffer_load_dwordx2 %0, %4, %5, 0 offen\n buffer_load_dwordx2 %1, %4, %5, 0 offen offset:16\n"
                             "buffer_load_dwordx2 %2, %4, %5, 0 offen offset:32\n buffer_load_dwordx2 %3, %4, %5, 0 offen offset:48\n"
                             "s_waitcnt vmcnt(4)\n"
                             : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3) : "v"(off), "s"(rsrc) : "memory");
            }
        } else if (KIND == 18 || KIND == 21) {
#pragma unroll
            for (int rr = 0; rr < 8; rr++) {
                rnd = rnd * 1664525u + 1013904223u;
                const uint32_t off = ((rnd >> 7) % n_records) * 64u;
                asm volatile("buffer_load_dwordx3 %0, %4, %5, 0 offen\n buffer_load_dwordx3 %1, %4, %5, 0 offen offset:16\n"
                             "buffer_load_dwordx3 %2, %4, %5, 0 offen offset:32\n buffer_load_dwordx3 %3, %4, %5, 0 offen offset:48\n"
                             "s_waitcnt vmcnt(4)\n"
                             : "=&v"(j0), "=&v"(j1), "=&v"(j2), "=&v"(j3) : "v"(off), "s"(rsrc) : "memory");
            }
        } else if (KIND == 22) {
#pragma unroll
            for (int rr = 0; rr < 8; rr++) {
                rnd = rnd * 1664525u + 1013904223u;
                const uint32_t off = ((rnd >> 7) % n_records) * 64u;
                asm volatile("buffer_load_dwordx4 %0, %4, %5, 0 offen\n buffer_load_dwordx4 %1, %4, %5, 0 offen offset:16\n"
                             "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:32\n buffer_load_dword %3, %4, %5, 0 offen offset:48\n"
                             "s_waitcnt vmcnt(4)\n"
                             : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(h3) : "v"(off), "s"(rsrc) : "memory");
            }
        } else if (KIND >= 23 && KIND <= 46) {
#define SRT_CAL8(INS) REP4(asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "s"(sk), "s"(smask) : "vcc");)
#define I23(n) "v_add_f32_e64 %" #n ", %" #n ", %8\n"
#define I24(n) "v_add_f32 %" #n ", 0x3f800001, %" #n "\n"
#define I25(n) "v_fmac_f32 %" #n ", %8, %8\n"
#define I26(n) "v_mov_b32 %" #n ", %8\n"
#define I27(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, %10\n"
#define I28(n) "v_cndmask_b32_e32 %" #n ", %" #n ", %8, vcc\n"
#define I29(n) "v_mul_f32 %" #n ", %9, %" #n "\n"
#define I30(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I32(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I33(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define I34(n) "v_add_f32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define I35(n) "v_fmamk_f32 %" #n ", %" #n ", 0x3f800001, %8\n"
#define I36(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define I37(n) "v_cmp_lt_f32_e64 s[90:91], %" #n ", %8\n"
#define I39(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc\n"
#define I40(n) "v_add_u32 %" #n ", %9, %" #n "\n"
            if (KIND == 23) { SRT_CAL8(I23) } else if (KIND == 24) { SRT_CAL8(I24) } else if (KIND == 25) { SRT_CAL8(I25) }
            else if (KIND == 26) { SRT_CAL8(I26) } else if (KIND == 27) { SRT_CAL8(I27) } else if (KIND == 28) { SRT_CAL8(I28) }
            else if (KIND == 29) { SRT_CAL8(I29) } else if (KIND == 30) { SRT_CAL8(I30) } else if (KIND == 32) { SRT_CAL8(I32) }
            else if (KIND == 33) { SRT_CAL8(I33) } else if (KIND == 34) { SRT_CAL8(I34) } else if (KIND == 35) { SRT_CAL8(I35) }
            else if (KIND == 36) { SRT_CAL8(I36) } else if (KIND == 39) { SRT_CAL8(I39) } else if (KIND == 40) { SRT_CAL8(I40) }
            else if (KIND == 37) {
                REP4(asm volatile(I37(0) I37(1) I37(2) I37(3) I37(4) I37(5) I37(6) I37(7)
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "s"(sk), "s"(smask) : "s90", "s91");)
            } else if (KIND == 38) {
                REP4(asm volatile("s_mov_b64 vcc, %10\n" I28(0) I28(1) I28(2) I28(3) I28(4) I28(5) I28(6) I28(7)
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "s"(sk), "s"(smask) : "vcc");)
            }
            else if (KIND >= 41 && KIND <= 46) {
#define A(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define SRT_CALX(BODY) REP4(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "s"(sk), "s"(smask) : "vcc");)
                if (KIND == 41) { SRT_CALX(I28(0) A(1) I28(2) A(3) I28(4) A(5) I28(6) A(7)) }                                 // stale VCC, every other instruction
                else if (KIND == 42) { SRT_CALX(I36(0) I28(1) I28(2) A(3) I36(4) I28(5) I28(6) A(7)) }                        // v_cmp -> two VOP2 consumers
                else if (KIND == 43) { SRT_CALX(I36(0) A(1) A(2) A(3) I28(4) A(5) A(6) A(7)) }                                // v_cmp ... three instructions ... consumer
                else if (KIND == 44) { SRT_CALX("s_and_b64 vcc, vcc, %10\n" I28(0) I28(1) A(2) A(3) A(4) A(5) A(6) A(7)) }      // SALU-written VCC, two VOP2 consumers (the INNER visit's tail)
                else if (KIND == 45) { SRT_CALX("s_and_b64 vcc, vcc, %10\n" I39(0) I39(1) A(2) A(3) A(4) A(5) A(6) A(7)) }      // the same with VOP3-encoded selects
                else { SRT_CALX(I28(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7)) }                                                  // one stale-VCC VOP2 select in eight
            }
            else if (KIND == 31) {
                REP4(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                                  "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(k2));)
            }
        } else if (KIND == 9) {
            REP4(asm volatile("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n"
                              "v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if (KIND >= 11) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (KIND == 10 || (KIND >= 12 && KIND <= 15) || (KIND >= 19 && KIND <= 22)) asm volatile("s_mov_b64 exec, %0" ::"s"(saved_exec));
    const uint32_t gwave = (blockIdx.x * blockDim.x + tid) >> 6;
    if ((tid & 63u) == 0u) cycles[gwave] = t1 - t0;
    // keep every accumulator alive
    float acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y +
                (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) + (float)(g0.x + g1.y + g2.z + g3.w) +
                (float)(h0 + h1 + h2 + h3 + i0.x + i1.y + i2.x + i3.y + j0.x + j1.y + j2.z + j3.x);
    if (acc == 12345.678f) sink[tid] = acc;
}

template <int KIND>
static hipError_t run_kind(uint32_t n_blocks, uint32_t threads, uint32_t iters, float *sink, unsigned long long *cycles, const float4 *table, uint32_t n_records, hipStream_t st) {
    const size_t lds = 96 * 1024;      // more than half a CU's LDS: exactly one workgroup per CU, so waves / SIMD = threads / 256
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&calib_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((calib_kernel<KIND>), dim3(n_blocks), dim3(threads), lds, st, iters, sink, cycles, table, n_records);
    return hipGetLastError();
}

hipError_t launch_calib(int kind, uint32_t n_blocks, uint32_t threads, uint32_t iters, float *sink, unsigned long long *cycles, const float4 *table, uint32_t n_records, hipStream_t st) {
    switch (kind) {
    case 0: return run_kind<0>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 1: return run_kind<1>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 2: return run_kind<2>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 3: return run_kind<3>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 4: return run_kind<4>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 5: return run_kind<5>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 6: return run_kind<6>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 7: return run_kind<7>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 8: return run_kind<8>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 9: return run_kind<9>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 10: return run_kind<10>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 11: return run_kind<11>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 12: return run_kind<12>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 13: return run_kind<13>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 14: return run_kind<14>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 15: return run_kind<15>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 16: return run_kind<16>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 17: return run_kind<17>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 18: return run_kind<18>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 19: return run_kind<19>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 20: return run_kind<20>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 21: return run_kind<21>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 22: return run_kind<22>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 23: return run_kind<23>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 24: return run_kind<24>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 25: return run_kind<25>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 26: return run_kind<26>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 27: return run_kind<27>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 28: return run_kind<28>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 29: return run_kind<29>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 30: return run_kind<30>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 31: return run_kind<31>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 32: return run_kind<32>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 33: return run_kind<33>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 34: return run_kind<34>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 35: return run_kind<35>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 36: return run_kind<36>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 37: return run_kind<37>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 38: return run_kind<38>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 39: return run_kind<39>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 40: return run_kind<40>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 41: return run_kind<41>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 42: return run_kind<42>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 43: return run_kind<43>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 44: return run_kind<44>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 45: return run_kind<45>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    case 46: return run_kind<46>(n_blocks, threads, iters, sink, cycles, table, n_records, st);
    default: return hipErrorInvalidValue;
    }
}
int calib_kinds() { return kCalibKinds; }

}  // namespace srt
