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

typedef unsigned short bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even, NaN kept NaN (plain cast path lowers to v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f32_to_bf16(float f)
{
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
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

}  // namespace mmr
