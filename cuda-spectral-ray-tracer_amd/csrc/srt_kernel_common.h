// srt_kernel_common.h -- the LDS map of a render workgroup, the block of launch uniforms kept in LDS, compile-time knobs and small
// wave-level helpers of srt_kernels.hip (kept apart so that kernel experiments in a second translation unit can share them).
#pragma once
#include "srt_device.h"
#include "srt_internal.h"

namespace srt {

// LDS map of one workgroup (W waves): 256 B launch uniforms | colour matching rows (96 float4) | inner-record cache: three
// float4 planes + one (16-bit refs) or two u32 planes of n_cached entries | W traversal stacks, each
// stack_depth * 64 lanes * (2 or 4) B, lane-interleaved.
constexpr int kLdsUniF4 = 16;          // 256 B block of launch-uniform values that only the cold paths read (see LdsUniforms)
constexpr int kLdsCmfF4 = 96;
constexpr int kLdsTablesF4 = kLdsUniF4 + kLdsCmfF4;

// Launch-uniform values used only by the pixel-switch / camera-ray blocks.  Kept in LDS instead of SGPRs: the persistent
// loop has ~190 live scalars otherwise, and the allocator spilled 80 of them into VGPR lanes, putting dozens of
// v_readlane into every traversal step.  A uniform-address ds_read is a broadcast and is only paid in the cold blocks.
struct LdsUniforms {
    float du[3], dv[3], p00[3], center[3], disk_u[3], disk_v[3], defocus_angle;
    uint32_t width, height, offx, offy, tx, ty, bx, by, tiles_x, n_tiles, rank, world, spp, n_lanes;
    uint32_t n_rows;          // queue length in rows of 64 pixel slots (= tiles_local unless expensive tiles were split)
    uint32_t lane_limit;
    uint32_t rng[2], tile_out[2], tile_order[2], tile_cost[2], pixel_counter[2];
    uint32_t tile_group_stride;
    uint32_t first_row_taken;       // bit w: wave w of this workgroup has taken its assigned first row
    uint32_t n_assigned_slots;      // queue slots handed out by assignment (the first row of every wave): the shared counter starts behind them
    uint32_t prio_cost[2], prio_full;      // per-tile probe cost (read-only in the render launch) and cost_max * spp (as float bits)
};
static_assert(sizeof(LdsUniforms) <= kLdsUniF4 * 16, "uniform block too large");
typedef __attribute__((address_space(3))) LdsUniforms lds_uniforms;
__device__ __forceinline__ void split_ptr(const void *p, __attribute__((address_space(3))) uint32_t *dst) {
    const unsigned long long v = (unsigned long long)p;
    dst[0] = (uint32_t)v; dst[1] = (uint32_t)(v >> 32);
}
template <typename T>
__device__ __forceinline__ T *join_ptr(uint32_t lo, uint32_t hi) { return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo); }

constexpr size_t kLdsBudget = 160 * 1024;

#ifndef SRT_PRIO_MODE
#define SRT_PRIO_MODE 1      /* wave priorities by remaining chain length (least slack first), see render_kernel */
#endif
#ifndef SRT_ASSIGN_FIRST_ROW
#define SRT_ASSIGN_FIRST_ROW 1   /* the first queue row of every wave is assigned by cost band instead of raced for (render_kernel, S3) */
#endif
#ifndef SRT_PRIO_L2
#define SRT_PRIO_L2 0        /* 1: the priorities are also compiled into the variants whose inner tree is partly served by L2 */
#endif
#ifndef SRT_PRIO_T1
#define SRT_PRIO_T1 256      /* tiers: remaining chain > T / 1024 of the longest chain of the launch */
#define SRT_PRIO_T2 64
#define SRT_PRIO_T3 16
#endif

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block-linear index of chunk pixel (i, j): idx = ty*28+tx + 448*(by*gridDim.x+bx)  (rendering.cu:156-165)
__device__ __forceinline__ uint32_t block_linear_idx(uint32_t i, uint32_t j, uint32_t tx, uint32_t ty, uint32_t bx) {
    uint32_t gbx = i / tx, gby = j / ty;
    uint32_t lx = i - gbx * tx, ly = j - gby * ty;
    return ly * tx + lx + tx * ty * (gby * bx + gbx);
}

}  // namespace srt
