// Micro-benchmark: would an fp32x3 conv tap loop built for OCCUPANCY (small LDS tiles, several independent 4-wave workgroups per CU,
// single-buffered weight groups reloaded from L2 in front of every barrier interval) keep the matrix pipe busier than today's
// 8-wave / one-workgroup-per-CU loop with its 48 MFMAs per wave between two barriers (50 % MFMA-busy)?
// Model of one workgroup: 4 waves, 64 x 64 wave tile (16 accumulator tiles), per interval: [every thread loads NLOAD x 16 B of
// "weights" from a 27-interval global image (L2-resident) and writes them to LDS] -> barrier -> MPI MFMAs per wave with
// RPM ds_read_b128 per MFMA interleaved (fragments from a read-only "A tile" + the weight area) -> barrier.  No A restage, no epilogue.
// Variants: workgroups per CU = 1, 2, 3 (by LDS size), MPI = 48 / 72 / 144.
// Build: hipcc --offload-arch=gfx950 -O3 occ_conv_model.hip -o occ_conv_model ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MPI, int NLOAD>
__global__ void __launch_bounds__(256) kmodel(const uint4* __restrict__ wimg, const uint4* __restrict__ seed, float* __restrict__ out, int iters,
                                              int lds_pad_unused)
{
    extern __shared__ uint4 sm[];          // [A area 2048 x 16 B = 32 KB][W area NLOAD x 256 x 16 B]
    const int tid = threadIdx.x;
    for (int i = 0; i < 8; ++i) sm[i * 256 + tid] = seed[(i * 256 + tid) % 4096];
    uint4* sw = sm + 2048;
    __syncthreads();
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        // weight group of this interval: L2-resident image, every workgroup reads the same addresses (like the packed weights)
        uint4 w[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) w[i] = wimg[((it % 27) * NLOAD + i) * 256 + tid];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) sw[i * 256 + tid] = w[i];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < MPI / 16; ++g) {   // 16 MFMAs per group with 4 A + 4 B fragment reads (0.5 reads per MFMA)
            uint4 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = sm[((g + i + it) & 7) * 256 + tid];
                fb[i] = sw[((g + i) % NLOAD) * 256 + ((tid + 64 * i) & 255)];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[n]), __builtin_bit_cast(bf16x8, fa[m]),
                                                                      acc[m][n], 0, 0, 0);
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MPI, int NLOAD>
void run(const char* name, int wg_per_cu, const uint4* wimg, const uint4* seed, float* out)
{
    const int iters = 2000;
    // LDS per workgroup chosen so that exactly wg_per_cu fit into 160 KB
    const int lds = wg_per_cu == 1 ? 100 * 1024 : wg_per_cu == 2 ? 70 * 1024 : 52 * 1024;
    auto kern = kmodel<MPI, NLOAD>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 10; ++l) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, wimg, seed, out, iters, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    const double flops = 10.0 * blocks * 4 * (double)iters * MPI * 2.0 * 16 * 16 * 32;
    const double tf = flops / (best * 1e-3) / 1e12;
    printf("%-40s %d WG/CU x 4 waves, %3d MFMA per wave and barrier interval, %d x 4 KB weight reload: %6.0f TFLOP/s = %.1f %% of 2.5 PFLOP/s\n", name,
           wg_per_cu, MPI, NLOAD, tf, tf / 25.0);
}

int main()
{
    std::vector<uint4> h(4096 + 27 * 8 * 256);
    srand(1);
    for (auto& v : h) {
        unsigned w[4];
        for (int j = 0; j < 4; ++j) {
            unsigned lo = ((rand() & 1) << 15) | ((119 + rand() % 8) << 7) | (rand() & 127);
            unsigned hi = ((rand() & 1) << 15) | ((119 + rand() % 8) << 7) | (rand() & 127);
            w[j] = lo | (hi << 16);
        }
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    uint4* buf; float* out;
    hipMalloc(&buf, h.size() * sizeof(uint4));
    hipMalloc(&out, 256 * 3 * 256 * sizeof(float));
    hipMemcpy(buf, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    const uint4* seed = buf; const uint4* wimg = buf + 4096;
    for (int wg = 1; wg <= 3; ++wg) {
        run<48, 2>("like today's tap (48 MFMA, 8 KB)", wg, wimg, seed, out);
        run<80, 3>("3-tap group (80 MFMA, 12 KB)", wg, wimg, seed, out);
        run<144, 6>("6-tap group (144 MFMA, 24 KB)", wg, wimg, seed, out);
    }
    return 0;
}
