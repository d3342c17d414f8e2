// Micro-benchmark: issue cost of whole-wave DPP shifts (wave_shr:1) against row_shr:1 and plain VALU on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 dpp_rate.hip -o dpp_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL>
__device__ __forceinline__ float dpp_add(float s, float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), CTRL, 0xf, 0xf, true));
}

// MODE 0: v_add_f32 (plain), 1: wave_shr:1 fused add, 2: row_shr:1 fused add, 3: ds_bpermute shift + add
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters)
{
    float v[8], s[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.001f + i; s[i] = v[i]; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) s[i] = v[i] + s[i] * 0.999f;
                else if (MODE == 1) s[i] = dpp_add<0x138>(s[i], v[i]);
                else if (MODE == 2) s[i] = dpp_add<0x111>(s[i], v[i]);
                else s[i] = v[i] + __shfl_up(s[i], 1, 64);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float a = 0;
    for (int i = 0; i < 8; ++i) a += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int blocks = 256 * 2, iters = 2000;   // 2 blocks of 4 waves per CU = 2 waves per SIMD
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    std::vector<long long> h(blocks);
    const char* names[4] = {"v_fma (plain VALU)", "v_add_f32_dpp wave_shr:1", "v_add_f32_dpp row_shr:1", "ds_bpermute + add"};
    for (int m = 0; m < 4; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0; for (auto c : h) s += c;
        printf("%-28s %.2f memtime-ticks per wave-instruction (2 waves/SIMD, 8 independent chains)\n", names[m], s / blocks / (iters * 64.0));
    }
    return 0;
}
