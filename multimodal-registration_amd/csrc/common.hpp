// Shared helpers for the gfx950 kernels behind include/mmr.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmr.h"

namespace mmr {

void set_hip_error(hipError_t e);

inline int check_launch()
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error(e);
        return MMR_EHIP;
    }
    return MMR_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// memory-bound launches: cap the grid and grid-stride (guide: Guideline 11).
inline int stream_grid(int64_t n, int block, int max_blocks = 256 * 16)
{
    int64_t g = (n + block - 1) / block;
    if (g > max_blocks) g = max_blocks;
    if (g < 1) g = 1;
    return (int)g;
}

// Sampling step per axis of ne.utils.resize: output index i reads input coordinate i * step.
// MMR_RESIZE_ALIGN_CORNERS: linspace(0, old-1, new) -> step (old-1)/(new-1); MMR_RESIZE_ARANGE_OVER_F: arange(new)/f ->
// step 1/f with f = zoom (> 0) or, zoom == 0, new/old per axis; positions past old-1 clamp to the edge (interpn).
inline int resize_steps(int X, int Y, int Z, int Xo, int Yo, int Zo, int grid_mode, float zoom, float& stx, float& sty,
                        float& stz)
{
    if (grid_mode == MMR_RESIZE_ALIGN_CORNERS) {
        stx = (float)(X - 1) / (float)(Xo > 1 ? Xo - 1 : 1);
        sty = (float)(Y - 1) / (float)(Yo > 1 ? Yo - 1 : 1);
        stz = (float)(Z - 1) / (float)(Zo > 1 ? Zo - 1 : 1);
        return 0;
    }
    if (grid_mode != MMR_RESIZE_ARANGE_OVER_F || zoom < 0.f) return 1;
    stx = zoom > 0.f ? 1.0f / zoom : (float)X / (float)Xo;
    sty = zoom > 0.f ? 1.0f / zoom : (float)Y / (float)Yo;
    stz = zoom > 0.f ? 1.0f / zoom : (float)Z / (float)Zo;
    return 0;
}

typedef unsigned short bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even, NaN kept NaN (plain cast path lowers to v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f32_to_bf16(float f)
{
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}

// two fp32 -> packed bf16 pair (low half = a), round-to-nearest-even: ONE v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) __bf16 mmr_bf16x2;
typedef __attribute__((ext_vector_type(2))) float mmr_f32x2;
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b)
{
    const mmr_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mmr_bf16x2));
}

// Dice ratio top / bottom under the two upstream variants: MMR_DICE_DIVIDE_NO_NAN =
// tf.math.divide_no_nan (0 where bottom == 0), MMR_DICE_MAX_EPS = top / max(bottom, 1e-5) (older voxelmorph).
#define MMR_DICE_EPS 1e-5f
__device__ __forceinline__ float dice_ratio(float top, float bot, int mode)
{
    if (mode == MMR_DICE_MAX_EPS) return top / fmaxf(bot, MMR_DICE_EPS);
    return (bot != 0.f) ? top / bot : 0.f;
}
// d ratio / d p through bottom (ga, same for every voxel) and through top per unit of y_true (gb): the
// gradient w.r.t. y_pred[v] is gb * y_true[v] + ga   (top = 2 sum t p, bottom = sum t + p)
__device__ __forceinline__ void dice_ratio_grad(float top, float bot, int mode, float& ga, float& gb)
{
    if (mode == MMR_DICE_MAX_EPS) {
        const bool clamped = !(bot > MMR_DICE_EPS);
        const float d = clamped ? MMR_DICE_EPS : bot;
        ga = clamped ? 0.f : -top / (d * d);
        gb = 2.f / d;
    } else if (bot != 0.f) {
        ga = -top / (bot * bot);
        gb = 2.f / bot;
    } else {
        ga = gb = 0.f;
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Tile of workgroup `bx` in pass `it` of a persistent grid of G workgroups over nt tiles, or -1.  Workgroups are dealt round-robin
// over the 8 XCDs (bx & 7), each with its own L2; the plain walk `tile = it * G + bx` therefore puts NEIGHBOURING tiles, which
// share their halo rows, on different XCDs and every halo read misses L2.  Here every XCD takes a contiguous run of the pass's
// tiles (G % 8 == 0; otherwise the plain walk).  MMR_NO_XCD_TILES: the plain walk, for same-box A/B builds.
__device__ __forceinline__ int xcd_tile(int bx, int G, int it, int nt)
{
    const int base = it * G;
    const int cnt = nt - base < G ? nt - base : G;
    if (cnt <= 0) return -1;
#ifndef MMR_NO_XCD_TILES
    if ((G & 7) == 0) {
        const int x = bx & 7, k = bx >> 3, qd = cnt >> 3, rm = cnt & 7;
        if (k >= qd + (x < rm ? 1 : 0)) return -1;
        return base + (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd) + k;
    }
#endif
    return bx < cnt ? base + bx : -1;
}

}  // namespace mmr
