// Micro-benchmark: what bf16 MFMA rate the chip SUSTAINS on random operands (it lowers its clock under matrix-core load:
// MI355X_MICROARCH.md "DVFS give-back"), so that a kernel's fraction of the nominal 2.5 PFLOP/s can be read next to the
// fraction of what a bare MFMA loop reaches on the same box.  Two loops, both 2 waves per SIMD on all 256 CUs:
//   reg : v_mfma_f32_16x16x32_bf16 back to back, operands fixed in registers (128 independent accumulator registers per wave)
//   lds : the same MFMAs with every operand pair re-read from LDS by ds_read_b128 (24 reads per 64 MFMAs, the ratio of the
//         conv kernel's 128x64 wave tile), no barriers, no global traffic
// Build: hipcc --offload-arch=gfx950 -O3 mfma_ceiling.hip -o mfma_ceiling ; run on the GPU box (~3 s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// 16 waves per workgroup (4 per SIMD), 64 x 64 wave tiles: 16 reads per 32 MFMAs, 64 accumulator registers per wave
template <bool LDS>
__global__ void __launch_bounds__(1024) k16(const uint4* __restrict__ seed, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters)
{
    __shared__ uint4 sm[1024 * 8];
    const int tid = threadIdx.x;
    for (int i = 0; i < 8; ++i) sm[i * 1024 + tid] = seed[(i * 1024 + tid) % 4096];
    __syncthreads();
    uint4 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) { fa[i] = sm[i * 1024 + tid]; fb[i] = sm[(4 + i) * 1024 + tid]; }
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (LDS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { fa[i] = sm[((i + ks + it) % 8) * 1024 + tid]; fb[i] = sm[((4 + i + ks + it) % 8) * 1024 + tid]; }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[n]), __builtin_bit_cast(bf16x8, fa[m]),
                                                                      acc[m][n], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 1024 + tid] = s;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <bool LDS>
__global__ void __launch_bounds__(512, 2) k(const uint4* __restrict__ seed, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters)
{
    __shared__ uint4 sm[512 * 12];                       // 96 KB: 12 fragments per lane
    const int tid = threadIdx.x;
    for (int i = 0; i < 12; ++i) sm[i * 512 + tid] = seed[(i * 512 + tid) % 4096];
    __syncthreads();
    uint4 fa[8], fb[4];
    for (int i = 0; i < 8; ++i) fa[i] = sm[i * 512 + tid];
    for (int i = 0; i < 4; ++i) fb[i] = sm[(8 + i) * 512 + tid];
    f32x4 acc[8][4];
    for (int m = 0; m < 8; ++m)
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (LDS) {
#pragma unroll
                for (int i = 0; i < 8; ++i) fa[i] = sm[((i + ks + it) % 12) * 512 + tid];
#pragma unroll
                for (int i = 0; i < 4; ++i) fb[i] = sm[((8 + i + ks + it) % 12) * 512 + tid];
            }
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[n]), __builtin_bit_cast(bf16x8, fa[m]),
                                                                      acc[m][n], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int m = 0; m < 8; ++m)
        for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = 4000;   // default: one 8-wave workgroup per CU = 2 waves per SIMD
    std::vector<uint4> h(4096);
    srand(1);
    for (auto& v : h) {                                 // random bf16 in [-1, 1): sign, exponent 119..126, random mantissa
        unsigned w[4];
        for (int j = 0; j < 4; ++j) {
            unsigned lo = ((rand() & 1) << 15) | ((119 + rand() % 8) << 7) | (rand() & 127);
            unsigned hi = ((rand() & 1) << 15) | ((119 + rand() % 8) << 7) | (rand() & 127);
            w[j] = lo | (hi << 16);
        }
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    uint4* seed; float* out; unsigned long long* st;
    hipMalloc(&seed, h.size() * sizeof(uint4));
    hipMalloc(&out, blocks * 1024 * sizeof(float));
    hipMalloc(&st, blocks * 2 * sizeof(unsigned long long));
    hipMemcpy(seed, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    std::vector<unsigned long long> hs(blocks * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {             // ~1 s of back-to-back launches per mode: the clock settles
            hipEventRecord(e0);
            for (int l = 0; l < 20; ++l) {
                if (mode == 0) hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(512), 0, 0, seed, out, st, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(512), 0, 0, seed, out, st, iters);
                else if (mode == 2) hipLaunchKernelGGL(k16<false>, dim3(blocks), dim3(1024), 0, 0, seed, out, st, iters);
                else hipLaunchKernelGGL(k16<true>, dim3(blocks), dim3(1024), 0, 0, seed, out, st, iters);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2 && ms < best) best = ms;       // settled rounds only
        }
        hipMemcpy(hs.data(), st, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double cyc = 0, real = 0;
        for (int b = 0; b < blocks; ++b) { cyc += hs[2 * b]; real += hs[2 * b + 1]; }
        const double flops = 20.0 * blocks * (mode < 2 ? 8 : 16) /*waves*/ * (double)iters * 64 /*mfma*/ * 2.0 * 16 * 16 * 32;
        const double clock_ghz = cyc / real * 0.1;      // s_memrealtime ticks at 100 MHz
        const double tf = flops / (best * 1e-3) / 1e12;
        // cycles per MFMA and SIMD from the wall rate at the in-kernel clock (16 = the instruction's issue interval)
        printf("%s: %.0f TFLOP/s (%.1f %% of 2.5 PFLOP/s), in-kernel clock %.2f GHz, %.1f cycles per MFMA and SIMD\n",
               mode == 0 ? "registers, 2 waves/SIMD" : mode == 1 ? "lds-fed,   2 waves/SIMD" : mode == 2 ? "registers, 4 waves/SIMD" : "lds-fed,   4 waves/SIMD (64x64 wave tiles)", tf, tf / 2500 * 100, clock_ghz, 16384.0 * 1024 * clock_ghz * 1e9 / (tf * 1e12));
    }
    return 0;
}
