// Device-side pieces of the SynthMorph image generator (neurite
// labels_to_image / draw_perlin as called at train_synthmorph.py:57-64,288-291;
// stages listed in SURVEY.md Appendix A9/A10).  All HBM-bound elementwise or
// small-stencil work; random numbers come from a counter-based Philox4x32-10
// so every draw is reproducible from (seed, element index) and independent of
// launch geometry.
#include "common.hpp"

namespace mmr {

// ----------------------------- Philox4x32-10 ----------------------------- //
struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ __forceinline__ uint4 operator()(uint64_t ctr, uint32_t stream) const
    {
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = stream, c3 = 0x9E3779B9u;
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
            const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b;
            c1 = (uint32_t)p1;
            c3 = (uint32_t)p0;
            c0 = n0;
            c2 = n2;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

__device__ __forceinline__ float2 box_muller(uint32_t a, uint32_t b)
{
    const float r = sqrtf(-2.0f * __logf(u01(a)));
    float s, c;
    __sincosf(6.283185307179586f * u01(b), &s, &c);
    return make_float2(r * c, r * s);
}

// out[i] = mean + std * N(0,1) (normal != 0) or lo + (hi-lo) * U(0,1); 4 values per Philox call
__global__ void __launch_bounds__(256)
philox_fill_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint32_t stream, int normal, float a, float b)
{
    const Philox ph(seed);
    const int64_t n4 = (n + 3) / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 r = ph((uint64_t)i, stream);
        float v[4];
        if (normal) {
            const float2 p = box_muller(r.x, r.y), q = box_muller(r.z, r.w);
            v[0] = a + b * p.x; v[1] = a + b * p.y; v[2] = a + b * q.x; v[3] = a + b * q.y;
        } else {
            v[0] = a + (b - a) * u01(r.x); v[1] = a + (b - a) * u01(r.y);
            v[2] = a + (b - a) * u01(r.z); v[3] = a + (b - a) * u01(r.w);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i * 4 + k < n) out[i * 4 + k] = v[k];
    }
}

__global__ void __launch_bounds__(256)
lut_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const uint8_t* __restrict__ lut, int64_t n)
{
    __shared__ uint8_t s[256];
    s[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = s[in[i]];
}

// image[b,v] = mean[b,lab] + std[b,lab] * noise ; noise injected (noise != null) or Philox
__global__ void __launch_bounds__(256)
gmm_sample_kernel(const uint8_t* __restrict__ lab, const float* __restrict__ means, const float* __restrict__ stds,
                  const float* __restrict__ noise, float* __restrict__ out, int B, int64_t nvox, int L, uint64_t seed,
                  uint32_t stream)
{
    const Philox ph(seed);
    const int64_t total = (int64_t)B * nvox;
    const int64_t n4 = (total + 3) / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float z[4];
        if (noise) {
#pragma unroll
            for (int k = 0; k < 4; ++k) z[k] = (i * 4 + k < total) ? noise[i * 4 + k] : 0.f;
        } else {
            const uint4 r = ph((uint64_t)i, stream);
            const float2 p = box_muller(r.x, r.y), q = box_muller(r.z, r.w);
            z[0] = p.x; z[1] = p.y; z[2] = q.x; z[3] = q.y;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t e = i * 4 + k;
            if (e < total) {
                const int b = (int)(e / nvox);
                const int l = lab[e];
                out[e] = (l < L) ? means[b * L + l] + stds[b * L + l] * z[k] : 0.f;
            }
        }
    }
}

// 1-D convolution along `axis` with a per-item kernel [B][W] ('SAME', zero padding)
__global__ void __launch_bounds__(256)
blur_axis_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ kern, int B, int X,
                 int Y, int Z, int axis, int W)
{
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t total = (int64_t)B * nvox;
    const int R = W / 2;
    const int64_t stride = axis == 0 ? (int64_t)Y * Z : (axis == 1 ? Z : 1);
    const int len = axis == 0 ? X : (axis == 1 ? Y : Z);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / nvox);
        const int64_t r = i - (int64_t)b * nvox;
        const int pos = (int)((r / stride) % len);
        const float* k = kern + (int64_t)b * W;
        float acc = 0.f;
        for (int j = 0; j < W; ++j) {
            const int q = pos + j - R;
            if (q >= 0 && q < len) acc += k[j] * in[i + (int64_t)(j - R) * stride];
        }
        out[i] = acc;
    }
}

// x = clip(x * exp(bias), lo, hi) in place; per-item min / max partials -> part[b][blk][2]
__global__ void __launch_bounds__(256)
bias_clip_minmax_kernel(float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ part,
                        int64_t nvox, float lo, float hi, int nblk)
{
    __shared__ float smin[4], smax[4];
    const int b = blockIdx.y;
    float* p = x + (int64_t)b * nvox;
    const float* bb = bias ? bias + (int64_t)b * nvox : nullptr;
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (int64_t)nblk * blockDim.x) {
        float v = p[i];
        if (bb) v *= __expf(bb[i]);
        v = fminf(fmaxf(v, lo), hi);
        p[i] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_down(mn, o, 64));
        mx = fmaxf(mx, __shfl_down(mx, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mn = fminf(mn, smin[w]); mx = fmaxf(mx, smax[w]); }
        part[((int64_t)b * nblk + blockIdx.x) * 2] = mn;
        part[((int64_t)b * nblk + blockIdx.x) * 2 + 1] = mx;
    }
}

// x = ((x - min) / (max - min)) ** exp(gamma[b]) in place
__global__ void __launch_bounds__(256)
minmax_gamma_kernel(float* __restrict__ x, const float* __restrict__ part, const float* __restrict__ gamma,
                    int64_t nvox, int nblk)
{
    // every workgroup reduces the nblk (min, max) partials itself -- with all 256 threads: as a loop in thread 0 this prologue
    // was 2 nblk dependent loads in front of every workgroup's work, 61 us for a 16-MB volume (min / max are order-independent:
    // same bits)
    __shared__ float s_mn[4], s_mx[4];
    const int b = blockIdx.y;
    float mn = INFINITY, mx = -INFINITY;
    for (int k = threadIdx.x; k < nblk; k += 256) {
        const float2 pm = *reinterpret_cast<const float2*>(part + ((int64_t)b * nblk + k) * 2);
        mn = fminf(mn, pm.x);
        mx = fmaxf(mx, pm.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
    mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    const float rng = mx - mn;
    const float inv = rng > 0.f ? 1.0f / rng : 0.f;
    const float e = gamma ? expf(gamma[b]) : 1.0f;
    float* p = x + (int64_t)b * nvox;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = (p[i] - mn) * inv;
        p[i] = (gamma != nullptr) ? powf(v, e) : v;
    }
}

__global__ void __launch_bounds__(256)
onehot_kernel(const uint8_t* __restrict__ lab, float* __restrict__ out, int64_t n, int L)
{
    const int64_t total = n * L;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(i % L);
        out[i] = (lab[i / L] == l) ? 1.0f : 0.0f;
    }
}

// argmax over the channel axis -> uint8 label (first maximum wins, like tf.argmax)
__global__ void __launch_bounds__(256)
argmax_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int64_t n, int C)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float* p = x + i * C;
        float best = p[0];
        int bi = 0;
        for (int c = 1; c < C; ++c)
            if (p[c] > best) { best = p[c]; bi = c; }
        out[i] = (uint8_t)bi;
    }
}

__global__ void __launch_bounds__(256)
axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] += a * x[i];
}

constexpr int MM_BLOCKS = 512;

}  // namespace mmr

using namespace mmr;

extern "C" int mmr_philox_normal_f32(float* out, int64_t n, uint64_t seed, uint32_t stream_id, float mean, float std,
                                     void* stream)
{
    if (!out || n < 1) return MMR_EINVAL;
    hipLaunchKernelGGL(philox_fill_kernel, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), out, n,
                       seed, stream_id, 1, mean, std);
    return check_launch();
}

extern "C" int mmr_philox_uniform_f32(float* out, int64_t n, uint64_t seed, uint32_t stream_id, float lo, float hi,
                                      void* stream)
{
    if (!out || n < 1) return MMR_EINVAL;
    hipLaunchKernelGGL(philox_fill_kernel, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), out, n,
                       seed, stream_id, 0, lo, hi);
    return check_launch();
}

extern "C" int mmr_lut_u8(const uint8_t* in, uint8_t* out, const uint8_t* lut256, int64_t n, void* stream)
{
    if (!in || !out || !lut256 || n < 1) return MMR_EINVAL;
    hipLaunchKernelGGL(lut_u8_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), in, out, lut256, n);
    return check_launch();
}

extern "C" int mmr_gmm_sample_f32(const uint8_t* labels, const float* means, const float* stds, const float* noise,
                                  float* out, int B, int64_t nvox, int L, uint64_t seed, uint32_t stream_id,
                                  void* stream)
{
    if (!labels || !means || !stds || !out || B < 1 || nvox < 1 || L < 1 || L > 256) return MMR_EINVAL;
    hipLaunchKernelGGL(gmm_sample_kernel, dim3(stream_grid((B * nvox + 3) / 4, 256)), dim3(256), 0, as_stream(stream),
                       labels, means, stds, noise, out, B, nvox, L, seed, stream_id);
    return check_launch();
}

extern "C" int mmr_blur_axis_f32(const float* in, float* out, const float* kern, int B, int X, int Y, int Z, int axis,
                                 int W, void* stream)
{
    if (!in || !out || !kern || in == out || B < 1 || X < 1 || Y < 1 || Z < 1 || axis < 0 || axis > 2 || W < 1 ||
        !(W & 1))
        return MMR_EINVAL;
    hipLaunchKernelGGL(blur_axis_kernel, dim3(stream_grid((int64_t)B * X * Y * Z, 256)), dim3(256), 0,
                       as_stream(stream), in, out, kern, B, X, Y, Z, axis, W);
    return check_launch();
}

extern "C" int64_t mmr_intensity_ws_bytes(int B) { return B < 1 ? MMR_EINVAL : (int64_t)B * MM_BLOCKS * 2 * sizeof(float); }

// x <- ((clip(x * exp(bias), lo, hi) - min_b) / (max_b - min_b)) ** exp(gamma[b]); bias / gamma optional
extern "C" int mmr_bias_clip_norm_gamma_f32(float* x, const float* bias, const float* gamma, void* ws, int B,
                                            int64_t nvox, float lo, float hi, void* stream)
{
    if (!x || !ws || B < 1 || nvox < 1) return MMR_EINVAL;
    int nblk = (int)((nvox + 256 * 8 - 1) / (256 * 8));
    if (nblk > MM_BLOCKS) nblk = MM_BLOCKS;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(bias_clip_minmax_kernel, dim3(nblk, B), dim3(256), 0, as_stream(stream), x, bias, (float*)ws,
                       nvox, lo, hi, nblk);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(minmax_gamma_kernel, dim3(nblk, B), dim3(256), 0, as_stream(stream), x, (const float*)ws, gamma,
                       nvox, nblk);
    return check_launch();
}

extern "C" int mmr_onehot_f32(const uint8_t* labels, float* out, int64_t n, int L, void* stream)
{
    if (!labels || !out || n < 1 || L < 1) return MMR_EINVAL;
    hipLaunchKernelGGL(onehot_kernel, dim3(stream_grid(n * L, 256)), dim3(256), 0, as_stream(stream), labels, out, n, L);
    return check_launch();
}

extern "C" int mmr_argmax_u8(const float* x, uint8_t* out, int64_t n, int C, void* stream)
{
    if (!x || !out || n < 1 || C < 1 || C > 256) return MMR_EINVAL;
    hipLaunchKernelGGL(argmax_u8_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), x, out, n, C);
    return check_launch();
}

extern "C" int mmr_axpy_f32(float* y, const float* x, float a, int64_t n, void* stream)
{
    if (!y || !x || n < 1) return MMR_EINVAL;
    hipLaunchKernelGGL(axpy_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), y, x, a, n);
    return check_launch();
}
