// Calibration of rocprofv3's FETCH_SIZE for the NCC kernel's load instruction (raw_buffer_load_b128, 16 B per lane, hardware
// bounds check): the kernels below read a KNOWN number of bytes exactly once each, so FETCH_SIZE (KiB) x 1024 / bytes says
// whether the guide's "x 2 on gfx950 for 16-B-per-lane streams" applies to it (raw ratio 0.5 -> yes).
//   stream_b128   : every lane one 16-B buffer load per trip, 268 MB swept once              (the NCC kernel's instruction)
//   stream_dword  : the same bytes as 4-B global loads                                       (control: ratio 1.0 expected)
//   reread_b128   : a 32-MB window read 8 times (256 MB of requests, 32 MB unique: re-reads from L2 / Infinity Cache)
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib ; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) stream_b128(const float* p, float* out, unsigned long long n16, unsigned bytes)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
    f4_t a = {0, 0, 0, 0};
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256)
        a += __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(i * 16), 0, 0));
    if (a.x + a.y + a.z + a.w == 12345.678f) out[0] = a.x;
}

__global__ void __launch_bounds__(256) stream_dword(const float* p, float* out, unsigned long long n4)
{
    float a = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (unsigned long long)gridDim.x * 256) a += p[i];
    if (a == 12345.678f) out[0] = a;
}

__global__ void __launch_bounds__(256) reread_b128(const float* p, float* out, unsigned long long n16, unsigned bytes, int reps)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
    f4_t a = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256)
            a += __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(i * 16), 0, 0));
    if (a.x + a.y + a.z + a.w == 12345.678f) out[0] = a.x;
}

int main()
{
    const size_t bytes = 256ull << 20;
    float *p, *out, *flush;
    hipMalloc(&p, bytes);
    hipMalloc(&out, 64);
    hipMalloc(&flush, 1ull << 30);
    hipMemset(p, 0, bytes);
    for (int it = 0; it < 3; ++it) {
        hipMemset(flush, 1, 1ull << 30);                 // push p out of the 256-MiB Infinity Cache
        hipLaunchKernelGGL(stream_b128, dim3(2048), dim3(256), 0, 0, p, out, bytes / 16, (unsigned)bytes);
        hipMemset(flush, 2, 1ull << 30);
        hipLaunchKernelGGL(stream_dword, dim3(2048), dim3(256), 0, 0, p, out, bytes / 4);
        hipMemset(flush, 3, 1ull << 30);
        hipLaunchKernelGGL(reread_b128, dim3(2048), dim3(256), 0, 0, p, out, (32ull << 20) / 16, (unsigned)(32u << 20), 8);
    }
    hipDeviceSynchronize();
    printf("fetch_calib: stream_b128 / stream_dword read %zu bytes once; reread_b128 requests %zu bytes, %zu unique\n", bytes, bytes,
           (size_t)(32u << 20));
    return 0;
}
