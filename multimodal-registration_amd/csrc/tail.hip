// HBM-bound "tail" of VxmDense: spatial transformer, align-corners resize,
// compose / scaling-and-squaring.  One thread per output element, channels
// fastest so a wave's 8 corner gathers are contiguous channel runs (NDHWC).
// Arithmetic order follows neurite's interpn with FP
// contraction off, so results match the fp32 restatement operation for
// operation.
#include "common.hpp"

#pragma clang fp contract(off)

namespace mmr {

struct Axis {
    int i0, i1;
    float w0, w1;
};

__device__ __forceinline__ Axis axis_setup(float loc, int maxi)
{
    const float m = (float)maxi;
    const float fl = floorf(loc);
    const float cl = fminf(fmaxf(loc, 0.f), m);
    const float l0 = fminf(fmaxf(fl, 0.f), m);
    const float l1 = fminf(l0 + 1.f, m);
    Axis a;
    a.i0 = (int)l0;
    a.i1 = (int)l1;
    a.w0 = l1 - cl;
    a.w1 = 1.f - a.w0;
    return a;
}

// sum over the 8 corners in itertools.product order, weight = (wx*wy)*wz.
// `p` points at channel c of voxel (0,0,0) of this batch item; strides in elements.
template <typename LoadF>
__device__ __forceinline__ float trilinear(const Axis& ax, const Axis& ay, const Axis& az,
                                           int64_t sx, int64_t sy, int64_t sz, LoadF ld)
{
    const int64_t x0 = ax.i0 * sx, x1 = ax.i1 * sx;
    const int64_t y0 = ay.i0 * sy, y1 = ay.i1 * sy;
    const int64_t z0 = az.i0 * sz, z1 = az.i1 * sz;
    // issue all 8 gathers before use
    const float v000 = ld(x0 + y0 + z0), v001 = ld(x0 + y0 + z1);
    const float v010 = ld(x0 + y1 + z0), v011 = ld(x0 + y1 + z1);
    const float v100 = ld(x1 + y0 + z0), v101 = ld(x1 + y0 + z1);
    const float v110 = ld(x1 + y1 + z0), v111 = ld(x1 + y1 + z1);
    const float w00 = ax.w0 * ay.w0, w01 = ax.w0 * ay.w1, w10 = ax.w1 * ay.w0, w11 = ax.w1 * ay.w1;
    float o = 0.f;
    o = o + (w00 * az.w0) * v000;
    o = o + (w00 * az.w1) * v001;
    o = o + (w01 * az.w0) * v010;
    o = o + (w01 * az.w1) * v011;
    o = o + (w10 * az.w0) * v100;
    o = o + (w10 * az.w1) * v101;
    o = o + (w11 * az.w0) * v110;
    o = o + (w11 * az.w1) * v111;
    return o;
}

__device__ __forceinline__ int nearest_idx(float loc, int maxi)
{
    // tf.round = round half to even; clamp after the int cast like interpn
    const float r = rintf(loc);
    const float c = fminf(fmaxf(r, 0.f), (float)maxi);
    return (int)c;
}

template <int INTERP, typename T>
__global__ void __launch_bounds__(256)
warp3d_kernel(const T* __restrict__ vol, const float* __restrict__ flow, T* __restrict__ out,
              int B, int X, int Y, int Z, int C, int has_fill, T fill, int channelwise)
{
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t total = (int64_t)B * nvox * C;
    const int64_t sz = C, sy = (int64_t)Z * C, sx = (int64_t)Y * Z * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t v = i / C;  // b*nvox + vox
        const int64_t b = v / nvox;
        const int64_t r = v - b * nvox;
        const int z = (int)(r % Z);
        const int y = (int)((r / Z) % Y);
        const int x = (int)(r / ((int64_t)Z * Y));
        const float* f = channelwise ? flow + (v * C + c) * 3 : flow + v * 3;
        const float lx = (float)x + f[0], ly = (float)y + f[1], lz = (float)z + f[2];
        const T* base = vol + b * nvox * C + c;
        T o;
        if (INTERP == MMR_INTERP_NEAREST) {
            const int ix = nearest_idx(lx, X - 1), iy = nearest_idx(ly, Y - 1), iz = nearest_idx(lz, Z - 1);
            o = base[ix * sx + iy * sy + iz * sz];
        } else {
            const Axis ax = axis_setup(lx, X - 1), ay = axis_setup(ly, Y - 1), az = axis_setup(lz, Z - 1);
            o = (T)trilinear(ax, ay, az, sx, sy, sz, [&](int64_t off) { return (float)base[off]; });
        }
        if (has_fill) {
            const bool oob = (lx < 0.f) | (lx > (float)(X - 1)) | (ly < 0.f) | (ly > (float)(Y - 1)) |
                             (lz < 0.f) | (lz > (float)(Z - 1));
            if (oob) o = fill;
        }
        out[i] = o;
    }
}

// One thread per OUTPUT VOXEL (axis setup and the 8 corner offsets are computed once, then the channels
// are looped) with 32-bit indexing inside a batch item; NC > 0 unrolls the channel loop.
template <int NC>
__global__ void __launch_bounds__(256)
resize_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int X, int Y, int Z, int C,
              int Xo, int Yo, int Zo, float stx, float sty, float stz, float mul, int pre_scale)
{
    const int nvo = Xo * Yo * Zo;
    const int64_t total = (int64_t)B * nvo;
    const int sz = C, sy = Z * C, sx = Y * Z * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / nvo);
        const int r = (int)(i - (int64_t)b * nvo);
        const int z = r % Zo;
        const int y = (r / Zo) % Yo;
        const int x = r / (Zo * Yo);
        const Axis ax = axis_setup((float)x * stx, X - 1);
        const Axis ay = axis_setup((float)y * sty, Y - 1);
        const Axis az = axis_setup((float)z * stz, Z - 1);
        const float* base = in + (int64_t)b * X * sx;
        float* o = out + i * C;
        const int x0 = ax.i0 * sx, x1 = ax.i1 * sx, y0 = ay.i0 * sy, y1 = ay.i1 * sy, z0 = az.i0 * sz, z1 = az.i1 * sz;
        const int offs[8] = {x0 + y0 + z0, x0 + y0 + z1, x0 + y1 + z0, x0 + y1 + z1,
                             x1 + y0 + z0, x1 + y0 + z1, x1 + y1 + z0, x1 + y1 + z1};
        const float w00 = ax.w0 * ay.w0, w01 = ax.w0 * ay.w1, w10 = ax.w1 * ay.w0, w11 = ax.w1 * ay.w1;
        const float wt[8] = {w00 * az.w0, w00 * az.w1, w01 * az.w0, w01 * az.w1,
                             w10 * az.w0, w10 * az.w1, w11 * az.w0, w11 * az.w1};
        const int nc = NC > 0 ? NC : C;
#pragma unroll
        for (int c = 0; c < nc; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = base[offs[k] + c];
                acc = acc + wt[k] * (pre_scale ? v * mul : v);
            }
            o[c] = pre_scale ? acc : acc * mul;
        }
    }
}

// out = s*b + (s*a) o (id + s*b), 3-channel fields; s folds VecInt's 1/2^n.
__global__ void __launch_bounds__(256)
compose_kernel(const float* __restrict__ a, const float* __restrict__ bf, float* __restrict__ out,
               int B, int X, int Y, int Z, float s)
{
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t total = (int64_t)B * nvox * 3;
    const int64_t sz = 3, sy = (int64_t)Z * 3, sx = (int64_t)Y * Z * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const int64_t v = i / 3;
        const int64_t b = v / nvox;
        const int64_t r = v - b * nvox;
        const int z = (int)(r % Z);
        const int y = (int)((r / Z) % Y);
        const int x = (int)(r / ((int64_t)Z * Y));
        const float* f = bf + v * 3;
        const float f0 = f[0] * s, f1 = f[1] * s, f2 = f[2] * s;
        const Axis ax = axis_setup((float)x + f0, X - 1);
        const Axis ay = axis_setup((float)y + f1, Y - 1);
        const Axis az = axis_setup((float)z + f2, Z - 1);
        const float* base = a + b * nvox * 3 + c;
        const float w = trilinear(ax, ay, az, sx, sy, sz, [&](int64_t off) { return base[off] * s; });
        const float own = (c == 0) ? f0 : (c == 1 ? f1 : f2);
        out[i] = own + w;
    }
}

// The same composition with ONE thread per voxel (the kernel above spends a thread per voxel AND channel: three axis set-ups, three
// times the 64-bit index arithmetic and 24 four-byte gathers for what is one set-up and eight 12-byte gathers).  Per channel the
// arithmetic is the expression of `trilinear` term for term, so the result is bit-identical.  blockIdx.y = batch item; 32-bit
// indexing inside an item (the host checks 3 nvox < 2^31).  VecInt's five squaring steps: 78 -> 39 us at 80^3, 86 -> 45 us at
// 80 x 80 x 96, 539 -> 212 us at 160 x 160 x 192, same box, same bits (profiles/r05_ab_compose.log).
struct F3v { float x, y, z; };
__global__ void __launch_bounds__(256)
compose3_kernel(const float* __restrict__ a, const float* __restrict__ bf, float* __restrict__ out, int X, int Y, int Z, float s)
{
    const int nvox = X * Y * Z;
    const int sy = Z * 3, sx = Y * Z * 3;
    const size_t item = (size_t)blockIdx.y * nvox * 3;
    const F3v* av = reinterpret_cast<const F3v*>(a + item);
    const F3v* bv = reinterpret_cast<const F3v*>(bf + item);
    F3v* ov = reinterpret_cast<F3v*>(out + item);
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nvox; r += gridDim.x * blockDim.x) {
        const int z = r % Z, q = r / Z;
        const int y = q % Y, x = q / Y;
        const F3v f = bv[r];
        const float f0 = f.x * s, f1 = f.y * s, f2 = f.z * s;
        const Axis ax = axis_setup((float)x + f0, X - 1);
        const Axis ay = axis_setup((float)y + f1, Y - 1);
        const Axis az = axis_setup((float)z + f2, Z - 1);
        const int x0 = ax.i0 * sx, x1 = ax.i1 * sx, y0 = ay.i0 * sy, y1 = ay.i1 * sy, z0 = az.i0 * 3, z1 = az.i1 * 3;
        const float* base = a + item;
        auto ld = [&](int off) { return *reinterpret_cast<const F3v*>(base + off); };
        // all 8 gathers before use
        const F3v v000 = ld(x0 + y0 + z0), v001 = ld(x0 + y0 + z1), v010 = ld(x0 + y1 + z0), v011 = ld(x0 + y1 + z1);
        const F3v v100 = ld(x1 + y0 + z0), v101 = ld(x1 + y0 + z1), v110 = ld(x1 + y1 + z0), v111 = ld(x1 + y1 + z1);
        const float w00 = ax.w0 * ay.w0, w01 = ax.w0 * ay.w1, w10 = ax.w1 * ay.w0, w11 = ax.w1 * ay.w1;
        const float k000 = w00 * az.w0, k001 = w00 * az.w1, k010 = w01 * az.w0, k011 = w01 * az.w1;
        const float k100 = w10 * az.w0, k101 = w10 * az.w1, k110 = w11 * az.w0, k111 = w11 * az.w1;
        auto mix = [&](float c000, float c001, float c010, float c011, float c100, float c101, float c110, float c111) {
            float o = 0.f;
            o = o + k000 * (c000 * s);
            o = o + k001 * (c001 * s);
            o = o + k010 * (c010 * s);
            o = o + k011 * (c011 * s);
            o = o + k100 * (c100 * s);
            o = o + k101 * (c101 * s);
            o = o + k110 * (c110 * s);
            o = o + k111 * (c111 * s);
            return o;
        };
        F3v o;
        o.x = f0 + mix(v000.x, v001.x, v010.x, v011.x, v100.x, v101.x, v110.x, v111.x);
        o.y = f1 + mix(v000.y, v001.y, v010.y, v011.y, v100.y, v101.y, v110.y, v111.y);
        o.z = f2 + mix(v000.z, v001.z, v010.z, v011.z, v100.z, v101.z, v110.z, v111.z);
        ov[r] = o;
    }
}

// one launch of the composition: per-voxel threads when the item's indices fit 32 bits, else the per-element kernel
static inline void launch_compose(const float* a, const float* b, float* out, int B, int X, int Y, int Z, float s, hipStream_t st)
{
    const int64_t nvox = (int64_t)X * Y * Z;
    if (nvox * 3 <= 0x7fffffffll && B <= 65535) {
        int gx = (int)((nvox + 255) / 256);
        if (gx > 65536) gx = 65536;
        hipLaunchKernelGGL(compose3_kernel, dim3(gx, B), dim3(256), 0, st, a, b, out, X, Y, Z, s);
    } else {
        hipLaunchKernelGGL(compose_kernel, dim3(stream_grid((int64_t)B * nvox * 3, 256)), dim3(256), 0, st, a, b, out, B, X, Y, Z, s);
    }
}

}  // namespace mmr

using namespace mmr;

extern "C" int mmr_warp3d_f32(const float* vol, const float* flow, float* out, int B, int X, int Y, int Z, int C,
                              int interp, int has_fill, float fill, int channelwise, void* stream)
{
    if (!vol || !flow || !out || B < 1 || X < 1 || Y < 1 || Z < 1 || C < 1) return MMR_EINVAL;
    if (interp != MMR_INTERP_LINEAR && interp != MMR_INTERP_NEAREST) return MMR_EINVAL;
    const int64_t total = (int64_t)B * X * Y * Z * C;
    const int grid = stream_grid(total, 256);
    if (interp == MMR_INTERP_LINEAR)
        hipLaunchKernelGGL((warp3d_kernel<MMR_INTERP_LINEAR, float>), dim3(grid), dim3(256), 0, as_stream(stream),
                           vol, flow, out, B, X, Y, Z, C, has_fill, fill, channelwise);
    else
        hipLaunchKernelGGL((warp3d_kernel<MMR_INTERP_NEAREST, float>), dim3(grid), dim3(256), 0, as_stream(stream),
                           vol, flow, out, B, X, Y, Z, C, has_fill, fill, channelwise);
    return check_launch();
}

extern "C" int mmr_warp3d_nearest_u8(const uint8_t* vol, const float* flow, uint8_t* out, int B, int X, int Y,
                                     int Z, int C, int has_fill, uint8_t fill, void* stream)
{
    if (!vol || !flow || !out || B < 1 || X < 1 || Y < 1 || Z < 1 || C < 1) return MMR_EINVAL;
    const int64_t total = (int64_t)B * X * Y * Z * C;
    hipLaunchKernelGGL((warp3d_kernel<MMR_INTERP_NEAREST, uint8_t>), dim3(stream_grid(total, 256)), dim3(256), 0,
                       as_stream(stream), vol, flow, out, B, X, Y, Z, C, has_fill, fill, 0);
    return check_launch();
}

extern "C" int mmr_resize_trilinear_f32(const float* in, float* out, int B, int X, int Y, int Z, int C, int Xo,
                                        int Yo, int Zo, float mul, int pre_scale, int grid_mode, float zoom, void* stream)
{
    if (!in || !out || B < 1 || X < 1 || Y < 1 || Z < 1 || C < 1 || Xo < 1 || Yo < 1 || Zo < 1) return MMR_EINVAL;
    float stx, sty, stz;
    if (resize_steps(X, Y, Z, Xo, Yo, Zo, grid_mode, zoom, stx, sty, stz)) return MMR_EINVAL;
    if ((int64_t)X * Y * Z * C > 0x7fffffff || (int64_t)Xo * Yo * Zo > 0x7fffffff) return MMR_EINVAL;
    const int64_t total = (int64_t)B * Xo * Yo * Zo;
    const dim3 grid(stream_grid(total, 256, 256 * 32)), blk(256);
    if (C == 1)
        hipLaunchKernelGGL(resize_kernel<1>, grid, blk, 0, as_stream(stream), in, out, B, X, Y, Z, C, Xo, Yo, Zo, stx,
                           sty, stz, mul, pre_scale);
    else if (C == 3)
        hipLaunchKernelGGL(resize_kernel<3>, grid, blk, 0, as_stream(stream), in, out, B, X, Y, Z, C, Xo, Yo, Zo, stx,
                           sty, stz, mul, pre_scale);
    else
        hipLaunchKernelGGL(resize_kernel<0>, grid, blk, 0, as_stream(stream), in, out, B, X, Y, Z, C, Xo, Yo, Zo, stx,
                           sty, stz, mul, pre_scale);
    return check_launch();
}

extern "C" int mmr_compose_f32(const float* a, const float* b, float* out, int B, int X, int Y, int Z, void* stream)
{
    if (!a || !b || !out || B < 1 || X < 1 || Y < 1 || Z < 1) return MMR_EINVAL;
    if (out == a || out == b) return MMR_EINVAL;
    launch_compose(a, b, out, B, X, Y, Z, 1.0f, as_stream(stream));
    return check_launch();
}

extern "C" int mmr_vecint_f32(const float* vel, float* out, float* tmp, int B, int X, int Y, int Z, int nsteps,
                              void* stream)
{
    if (!vel || !out || !tmp || B < 1 || X < 1 || Y < 1 || Z < 1 || nsteps < 0 || nsteps > 30) return MMR_EINVAL;
    if (out == vel || tmp == vel || out == tmp) return MMR_EINVAL;
    const int64_t total = (int64_t)B * X * Y * Z * 3;
    hipStream_t st = as_stream(stream);
    if (nsteps == 0) {
        if (hipMemcpyAsync(out, vel, total * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return MMR_EHIP;
        return MMR_OK;
    }
    // ping-pong so that the last step lands in `out`
    float* bufs[2] = {out, tmp};
    int cur = (nsteps % 2 == 1) ? 0 : 1;  // destination of step 0
    const float* src = vel;
    float s = 1.0f / (float)(1 << nsteps);
    for (int k = 0; k < nsteps; ++k) {
        float* dst = bufs[cur];
        launch_compose(src, src, dst, B, X, Y, Z, s, st);
        int rc = check_launch();
        if (rc) return rc;
        src = dst;
        s = 1.0f;
        cur ^= 1;
    }
    return MMR_OK;
}

// Forward scaling-and-squaring that keeps the input of every squaring step for the backward pass:
// steps[k-1] = output of step k-1 (k = 1..nsteps-1), out = output of the last step.
extern "C" int mmr_vecint_save_f32(const float* vel, float* steps, float* out, int B, int X, int Y, int Z, int nsteps,
                                   void* stream)
{
    if (!vel || !out || B < 1 || X < 1 || Y < 1 || Z < 1 || nsteps < 0 || nsteps > 30) return MMR_EINVAL;
    if (nsteps > 1 && !steps) return MMR_EINVAL;
    const int64_t total = (int64_t)B * X * Y * Z * 3;
    hipStream_t st = as_stream(stream);
    if (nsteps == 0) {
        if (hipMemcpyAsync(out, vel, total * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return MMR_EHIP;
        return MMR_OK;
    }
    const float* src = vel;
    float s = 1.0f / (float)(1 << nsteps);
    for (int k = 0; k < nsteps; ++k) {
        float* dst = (k == nsteps - 1) ? out : steps + (int64_t)k * total;
        launch_compose(src, src, dst, B, X, Y, Z, s, st);
        int rc = check_launch();
        if (rc) return rc;
        src = dst;
        s = 1.0f;
    }
    return MMR_OK;
}
