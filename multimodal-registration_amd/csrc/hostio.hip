// Host <-> device hand-over of whole volumes for `model.predict([moving, fixed])` (3d_reg.py:310-314: nibabel's
// get_fdata() float64 arrays in, NumPy arrays out).  The copy engines move pinned memory at PCIe speed, but what the
// caller holds is pageable float64: converting it on the host into torch's (coherent) pinned buffers ran at ~1 GB/s on
// some boxes of this pool and made predict() 2x the forward.  Here the host does no arithmetic at all:
//   * mmr_host_register pins the caller's own pages for the duration of the call, or mmr_host_alloc hands out
//     CPU-cached (non-coherent) pinned staging memory that a plain memcpy fills at memory speed;
//   * mmr_cast_to_f32 is a kernel that READS that host memory directly over PCIe (16 B per lane, grid-stride, enough
//     loads in flight to fill the link) and writes fp32 into HBM -- dtype conversion and H2D in one pass;
//   * mmr_copy_to_host is the same idea the other way round for the outputs.
#include "common.hpp"

namespace mmr {

// n elements of `T` at src (host-pinned or device) -> fp32 at dst.  Four elements per thread and trip.
// WIDE: src is 16-B aligned (double: two 16-B loads per trip; float: one); otherwise element loads.
template <typename T, bool WIDE>
__global__ void __launch_bounds__(256) cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            T v[4];
            if constexpr (WIDE && sizeof(T) == 8) {
                const double2 a = *reinterpret_cast<const double2*>(src + i);
                const double2 b = *reinterpret_cast<const double2*>(src + i + 2);
                v[0] = (T)a.x; v[1] = (T)a.y; v[2] = (T)b.x; v[3] = (T)b.y;
            } else if constexpr (WIDE && sizeof(T) == 4) {
                const float4 a = *reinterpret_cast<const float4*>(src + i);
                v[0] = (T)a.x; v[1] = (T)a.y; v[2] = (T)a.z; v[3] = (T)a.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = src[i + k];
            }
            *reinterpret_cast<float4*>(dst + i) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
        } else {
            for (int64_t k = i; k < n; ++k) dst[k] = (float)src[k];
        }
    }
}

// device -> pinned host, 16 B per lane (the tail element-wise)
__global__ void __launch_bounds__(256) copy_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t nw)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < nw; i += stride) {
        if (i + 3 < nw) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
        else for (int64_t k = i; k < nw; ++k) dst[k] = src[k];
    }
}

static int hip_rc(hipError_t e)
{
    if (e == hipSuccess) return MMR_OK;
    set_hip_error(e);
    (void)hipGetLastError();
    return MMR_EHIP;
}

}  // namespace mmr

using namespace mmr;

extern "C" int mmr_host_alloc(void** out, int64_t bytes, int cached)
{
    if (!out || bytes <= 0) return MMR_EINVAL;
    *out = nullptr;
    // cached != 0: hipHostMallocNonCoherent -- ordinary write-back pages on the CPU side (memcpy at memory speed); the GPU
    // sees a consistent view at kernel / copy boundaries, which is all a staging buffer needs
    return hip_rc(hipHostMalloc(out, (size_t)bytes, hipHostMallocPortable | hipHostMallocMapped |
                                                    (cached ? hipHostMallocNonCoherent : hipHostMallocCoherent)));
}

extern "C" int mmr_host_free(void* p)
{
    if (!p) return MMR_OK;
    return hip_rc(hipHostFree(p));
}

extern "C" int mmr_host_register(void* p, int64_t bytes, void** dev_ptr)
{
    if (!p || bytes <= 0 || !dev_ptr) return MMR_EINVAL;
    *dev_ptr = nullptr;
    const int rc = hip_rc(hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    if (rc != MMR_OK) return rc;
    const int rc2 = hip_rc(hipHostGetDevicePointer(dev_ptr, p, 0));
    if (rc2 != MMR_OK) (void)hipHostUnregister(p);
    return rc2;
}

extern "C" int mmr_host_unregister(void* p)
{
    if (!p) return MMR_EINVAL;
    return hip_rc(hipHostUnregister(p));
}

extern "C" int mmr_cast_to_f32(const void* src, float* dst, int64_t n, int src_dtype, void* stream)
{
    if (!src || !dst || n < 0) return MMR_EINVAL;
    if (n == 0) return MMR_OK;
    // enough 16/32-B loads in flight to cover a PCIe round trip when src is host memory: 2048 blocks x 256 threads
    const int grid = stream_grid((n + 3) / 4, 256, 2048);
    hipStream_t s = as_stream(stream);
    if ((uintptr_t)dst & 15) return MMR_EINVAL;
    const bool wide = ((uintptr_t)src & 15) == 0;
    switch (src_dtype) {
        case MMR_HOST_F64:
            if (wide) cast_to_f32_kernel<double, true><<<grid, 256, 0, s>>>(static_cast<const double*>(src), dst, n);
            else cast_to_f32_kernel<double, false><<<grid, 256, 0, s>>>(static_cast<const double*>(src), dst, n);
            break;
        case MMR_HOST_F32:
            if (wide) cast_to_f32_kernel<float, true><<<grid, 256, 0, s>>>(static_cast<const float*>(src), dst, n);
            else cast_to_f32_kernel<float, false><<<grid, 256, 0, s>>>(static_cast<const float*>(src), dst, n);
            break;
        case MMR_HOST_U8: cast_to_f32_kernel<uint8_t, false><<<grid, 256, 0, s>>>(static_cast<const uint8_t*>(src), dst, n); break;
        case MMR_HOST_I16: cast_to_f32_kernel<int16_t, false><<<grid, 256, 0, s>>>(static_cast<const int16_t*>(src), dst, n); break;
        default: return MMR_EINVAL;
    }
    return check_launch();
}

extern "C" int mmr_copy_to_host(const void* src_dev, void* dst_host, int64_t bytes, void* stream)
{
    if (!src_dev || !dst_host || bytes < 0 || (bytes & 3) || (((uintptr_t)src_dev | (uintptr_t)dst_host) & 15)) return MMR_EINVAL;
    if (bytes == 0) return MMR_OK;
    const int64_t nw = bytes / 4;
    const int grid = stream_grid((nw + 3) / 4, 256, 2048);
    copy_words_kernel<<<grid, 256, 0, as_stream(stream)>>>(static_cast<const uint32_t*>(src_dev),
                                                          static_cast<uint32_t*>(dst_host), nw);
    return check_launch();
}

// Copy-engine transfer between pinned (mmr_host_alloc / mmr_host_register) host memory and the device: no compute unit is
// involved, which matters beside kernels that own whole CUs (a PCIe-reading kernel on a side stream kept the 512-thread conv
// workgroups off every CU it sat on: predict() on a batch of pairs ran SLOWER overlapped than serial).  kind 0 = host -> device,
// 1 = device -> host.
extern "C" int mmr_memcpy_async(void* dst, const void* src, int64_t bytes, int kind, void* stream)
{
    if (!dst || !src || bytes < 0 || (kind != 0 && kind != 1)) return MMR_EINVAL;
    if (bytes == 0) return MMR_OK;
    return hip_rc(hipMemcpyAsync(dst, src, (size_t)bytes, kind == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, as_stream(stream)));
}
