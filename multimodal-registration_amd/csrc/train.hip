// Backward / training kernels for the SynthMorph step
// (train_synthmorph.py:296-308,335-344: VxmDense + SpatialTransformer + Dice +
// Grad('l2') under Adam).  Everything here is fp32.
//
// Design notes (MI355X-first, not a translation of TF's autodiff graph):
//  * Dice(one_hot(lab2), warp(one_hot(lab1), flow)) and its flow-gradient are
//    computed straight from the two uint8 label volumes: the L-channel one-hot
//    tensors (426 MB each at 160^3 x 26) and the warped prediction are never
//    materialised.  Reads per voxel: 2 B of labels + 12 B of flow instead of
//    ~210 B.  Results are identical to the one-hot formulation because a
//    trilinear blend of {0,1} indicators is the sum of the corner weights
//    whose corner carries that label.
//  * Gather-shaped adjoints (resize, scaling-and-squaring) use float atomics:
//    their volume is small (<= 100 M adds) next to the 1.3 TB/s atomic rate.
//  * wgrad is an MFMA GEMM over voxels (v_mfma_f32_32x32x2_f32) with
//    per-workgroup partial slabs reduced in a fixed order (reproducible).
#include <cstdlib>

#include "common.hpp"

namespace mmr {

constexpr int TB = 256;

__device__ __forceinline__ double block_sum_d(double v, double* sh)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) r += sh[i];
    return r;
}

struct AxisG {
    int i0, i1;
    float w0, w1, inr;  // inr: d(clipped)/d(loc) = 1 inside [0,max] (tf.clip_by_value gradient), else 0
};

__device__ __forceinline__ AxisG axis_setup_g(float loc, int maxi)
{
    const float m = (float)maxi;
    const float fl = floorf(loc);
    const float cl = fminf(fmaxf(loc, 0.f), m);
    const float l0 = fminf(fmaxf(fl, 0.f), m);
    const float l1 = fminf(l0 + 1.f, m);
    AxisG a;
    a.i0 = (int)l0;
    a.i1 = (int)l1;
    a.w0 = l1 - cl;
    a.w1 = 1.f - a.w0;
    a.inr = (loc >= 0.f && loc <= m) ? 1.f : 0.f;
    return a;
}

// voxel index -> (x, y, z): 32-bit divisions whenever the volume has fewer than 2^31 voxels (the 64-bit ones are ~100
// instructions each and these per-voxel kernels did three per voxel)
__device__ __forceinline__ void voxel_xyz(int64_t v, int Y, int Z, bool small, int& x, int& y, int& z)
{
    if (small) {
        const unsigned u = (unsigned)v, q = u / (unsigned)Z;
        z = (int)(u - q * (unsigned)Z);
        x = (int)(q / (unsigned)Y);
        y = (int)(q - (unsigned)x * (unsigned)Y);
    } else {
        z = (int)(v % Z);
        y = (int)((v / Z) % Y);
        x = (int)(v / ((int64_t)Y * Z));
    }
}

// ------------------------------------------------------------------------- //
// Dice from label maps                                                      //
// ------------------------------------------------------------------------- //
// part[b][blk][l][2] = (sum_v t*p, sum_v t+p) partial sums; per-thread private LDS columns -> ordered.
__global__ void __launch_bounds__(TB)
dice_labels_partial_kernel(const uint8_t* __restrict__ lab1, const uint8_t* __restrict__ lab2,
                           const float* __restrict__ flow, double* __restrict__ part, int X, int Y, int Z, int L,
                           int nblk, int zeropad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_tp = reinterpret_cast<float*>(smem);  // [L][TB]
    float* s_bt = s_tp + L * TB;                   // [L][TB]
    const int tid = threadIdx.x;
    for (int l = 0; l < L; ++l) { s_tp[l * TB + tid] = 0.f; s_bt[l * TB + tid] = 0.f; }
    const int b = blockIdx.y;
    const int64_t nvox = (int64_t)X * Y * Z;
    const uint8_t* a1 = lab1 + (int64_t)b * nvox;
    const uint8_t* a2 = lab2 + (int64_t)b * nvox;
    const float* f = flow + (int64_t)b * nvox * 3;
    const int64_t sy = Z, sx = (int64_t)Y * Z;
    const bool small = nvox < (int64_t)0x7fffffff;
    for (int64_t v = (int64_t)blockIdx.x * TB + tid; v < nvox; v += (int64_t)nblk * TB) {
        int x, y, z;
        voxel_xyz(v, Y, Z, small, x, y, z);
        const AxisG ax = axis_setup_g((float)x + f[v * 3], X - 1);
        const AxisG ay = axis_setup_g((float)y + f[v * 3 + 1], Y - 1);
        const AxisG az = axis_setup_g((float)z + f[v * 3 + 2], Z - 1);
        const int t = a2[v];
        float pt = 0.f;  // pred at the target's label
        float wl[8];
        int ll[8];
        float p0 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int cx = c >> 2, cy = (c >> 1) & 1, cz = c & 1;
            wl[c] = ((cx ? ax.w1 : ax.w0) * (cy ? ay.w1 : ay.w0)) * (cz ? az.w1 : az.w0);
            ll[c] = a1[(cx ? ax.i1 : ax.i0) * sx + (cy ? ay.i1 : ay.i0) * sy + (cz ? az.i1 : az.i0)];
            if (ll[c] == 0) p0 += wl[c];
        }
        // losses.py:34-54 intent: drop voxels whose label-0 channel is >= 1 in either map
        if (zeropad && (t == 0 || p0 >= 1.f)) continue;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float w = wl[c];
            const int l = ll[c];
            if (l < L) s_bt[l * TB + tid] += w;
            if (l == t) pt += w;
        }
        if (t < L) {
            s_tp[t * TB + tid] += pt;
            s_bt[t * TB + tid] += 1.f;
        }
    }
    __syncthreads();
    // reduction over the 256 private columns in a fixed order: a wave per label (the one-thread-per-label form walked 2 x 256 LDS
    // words serially with 26 of 256 threads busy: ~16 us per workgroup of a 74-us kernel)
    const int lane = tid & 63, wv = tid >> 6;
    for (int l = wv; l < L; l += TB / 64) {
        double st = 0.0, sb = 0.0;
#pragma unroll
        for (int k = 0; k < TB / 64; ++k) { st += (double)s_tp[l * TB + k * 64 + lane]; sb += (double)s_bt[l * TB + k * 64 + lane]; }
        st = wave_sum(st);
        sb = wave_sum(sb);
        if (lane == 0) {
            double* o = part + (((int64_t)b * nblk + blockIdx.x) * L + l) * 2;
            o[0] = st;
            o[1] = sb;
        }
    }
}

// top_bot[b][l] = (2*sum tp, sum t+p); loss[0] = -mean divide_no_nan(top, bot).
// One wave per (b,l): ordered strided partial sums + wave reduction (was a serial 1024-long loop per thread).
__global__ void __launch_bounds__(64)
dice_labels_sum_kernel(const double* __restrict__ part, float* __restrict__ top_bot, int L, int nblk)
{
    const int i = blockIdx.x;  // b * L + l
    const int b = i / L, l = i % L;
    double st = 0.0, sb = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) {
        const double* o = part + (((int64_t)b * nblk + k) * L + l) * 2;
        st += o[0];
        sb += o[1];
    }
    st = wave_sum(st);
    sb = wave_sum(sb);
    if (threadIdx.x == 0) {
        top_bot[i * 2] = (float)(2.0 * st);
        top_bot[i * 2 + 1] = (float)sb;
    }
}

__global__ void __launch_bounds__(TB)
dice_labels_final_kernel(const float* __restrict__ top_bot, float* __restrict__ loss, int B, int L, int zeropad,
                         int mode)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < B * L; i += blockDim.x) {
        const int b = i / L, l = i % L;
        const float ft = top_bot[i * 2], fb = top_bot[i * 2 + 1];
        const bool counted = !zeropad || (b == 0 && l >= 1);  // zeropad: labels 1..L-1 of batch item 0 only
        if (counted) acc += (double)dice_ratio(ft, fb, mode);
    }
    const double s = block_sum_d(acc, sh);
    if (threadIdx.x == 0) loss[0] = (float)(-s / (double)(zeropad ? (L - 1) : B * L));
}

// d_flow[b,v,:] (+)= scale * d(dice)/d(flow); dice = -1/(B L) sum_{b,l} top/bot
__global__ void __launch_bounds__(TB)
dice_labels_bwd_kernel(const uint8_t* __restrict__ lab1, const uint8_t* __restrict__ lab2,
                       const float* __restrict__ flow, const float* __restrict__ top_bot, float* __restrict__ dflow,
                       int B, int X, int Y, int Z, int L, float scale, int accumulate, int zeropad, int mode)
{
    __shared__ float sA[256], sB[256];  // G[l] = sA[l] + [l == target] * sB[l]
    const int b = blockIdx.y;
    const float c = zeropad ? ((b == 0) ? -scale / (float)(L - 1) : 0.f) : -scale / (float)(B * L);
    for (int l = threadIdx.x; l < 256; l += TB) {
        float a = 0.f, bb = 0.f;
        if (l < L && !(zeropad && l == 0)) {
            const float top = top_bot[(b * L + l) * 2], bot = top_bot[(b * L + l) * 2 + 1];
            float ga, gb;
            dice_ratio_grad(top, bot, mode, ga, gb);  // via bot: -top/bot^2; via top: 2 t / bot
            a = c * ga;
            bb = c * gb;
        }
        sA[l] = a;
        sB[l] = bb;
    }
    __syncthreads();
    const int64_t nvox = (int64_t)X * Y * Z;
    const uint8_t* a1 = lab1 + (int64_t)b * nvox;
    const uint8_t* a2 = lab2 + (int64_t)b * nvox;
    const float* f = flow + (int64_t)b * nvox * 3;
    float* df = dflow + (int64_t)b * nvox * 3;
    const int64_t sy = Z, sx = (int64_t)Y * Z;
    const bool small = nvox < (int64_t)0x7fffffff;
    for (int64_t v = (int64_t)blockIdx.x * TB + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * TB) {
        int x, y, z;
        voxel_xyz(v, Y, Z, small, x, y, z);
        const AxisG ax = axis_setup_g((float)x + f[v * 3], X - 1);
        const AxisG ay = axis_setup_g((float)y + f[v * 3 + 1], Y - 1);
        const AxisG az = axis_setup_g((float)z + f[v * 3 + 2], Z - 1);
        const int t = a2[v];
        float G[8];
        float p0 = 0.f;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int cx = cc >> 2, cy = (cc >> 1) & 1, cz = cc & 1;
            const int l = a1[(cx ? ax.i1 : ax.i0) * sx + (cy ? ay.i1 : ay.i0) * sy + (cz ? az.i1 : az.i0)];
            G[cc] = sA[l] + (l == t ? sB[l] : 0.f);
            if (l == 0) p0 += ((cx ? ax.w1 : ax.w0) * (cy ? ay.w1 : ay.w0)) * (cz ? az.w1 : az.w0);
        }
        if (zeropad && (t == 0 || p0 >= 1.f)) {  // masked voxel: no gradient (the mask itself is piecewise constant)
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) G[cc] = 0.f;
        }
        // d out / d loc_x = inr_x * sum_{cy,cz} wy wz (G[1,cy,cz] - G[0,cy,cz]) etc.
        const float gx = ax.inr * (ay.w0 * az.w0 * (G[4] - G[0]) + ay.w0 * az.w1 * (G[5] - G[1]) +
                                   ay.w1 * az.w0 * (G[6] - G[2]) + ay.w1 * az.w1 * (G[7] - G[3]));
        const float gy = ay.inr * (ax.w0 * az.w0 * (G[2] - G[0]) + ax.w0 * az.w1 * (G[3] - G[1]) +
                                   ax.w1 * az.w0 * (G[6] - G[4]) + ax.w1 * az.w1 * (G[7] - G[5]));
        const float gz = az.inr * (ax.w0 * ay.w0 * (G[1] - G[0]) + ax.w0 * ay.w1 * (G[3] - G[2]) +
                                   ax.w1 * ay.w0 * (G[5] - G[4]) + ax.w1 * ay.w1 * (G[7] - G[6]));
        if (accumulate) { df[v * 3] += gx; df[v * 3 + 1] += gy; df[v * 3 + 2] += gz; }
        else { df[v * 3] = gx; df[v * 3 + 1] = gy; df[v * 3 + 2] = gz; }
    }
}

// ------------------------------------------------------------------------- //
// Grad-l2 backward: dflow (+)= scale*loss_mult/3 * sum_d (2/n_d) * (diff_prev - diff_next)   //
// ------------------------------------------------------------------------- //
// a wave per (b, x, y) row of Z * C contiguous floats: no per-element 64-bit index arithmetic (see grad_l2_partial_kernel)
__global__ void __launch_bounds__(TB)
grad_l2_bwd_kernel(const float* __restrict__ f, float* __restrict__ df, int B, int X, int Y, int Z, int C,
                   float cx, float cy, float cz, int accumulate)
{
    const int rowlen = Z * C;
    const int64_t sy = rowlen, sx = (int64_t)Y * rowlen;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t nrows = (int64_t)B * X * Y;
    for (int64_t row = (int64_t)blockIdx.x * (TB / 64) + wv; row < nrows; row += (int64_t)gridDim.x * (TB / 64)) {
        const int y = (int)(row % Y), x = (int)((row / Y) % X);
        const float* r = f + row * rowlen;
        float* o = df + row * rowlen;
        const bool lx = x > 0, hx = x + 1 < X, ly = y > 0, hy = y + 1 < Y;
        for (int t = lane; t < rowlen; t += 64) {
            const float c0 = r[t];
            float g = 0.f;
            if (lx) g += cx * (c0 - r[t - sx]);
            if (hx) g -= cx * (r[t + sx] - c0);
            if (ly) g += cy * (c0 - r[t - sy]);
            if (hy) g -= cy * (r[t + sy] - c0);
            if (t >= C) g += cz * (c0 - r[t - C]);
            if (t + C < rowlen) g -= cz * (r[t + C] - c0);
            if (accumulate) o[t] += g; else o[t] = g;
        }
    }
}

// ------------------------------------------------------------------------- //
// adjoint of the align-corners resize: d_in[j] = sum_i w_i(j) * mul * d_out[i]
// Gather form: the sampling positions i * st are a regular lattice, so the outputs whose two corners touch input
// index j are a short index range per axis (about 2 / st of them); no atomics -> one pass, fixed summation order.
// The weights are recomputed exactly as the forward computes them (axis_setup_g of (float)i * st).
// ------------------------------------------------------------------------- //
__device__ __forceinline__ void axis_range(int j, float st, int no, int maxi, int& lo, int& hi)
{
    if (st > 0.f) {
        // output i reads inputs floor(i st) and floor(i st) + 1: it touches j iff i st lies in [j - 1, j + 1).  No extra margin:
        // an index that floor / ceil drop through rounding sits on the edge of that interval, where its weight is 0 to rounding
        // (the margins of +-1 this replaces made 7 - 8 candidates per axis for the x2 resize instead of 5 - 6: 2.5x the gather)
        lo = (int)floorf((float)(j - 1) / st);
        hi = (int)ceilf((float)(j + 1) / st);
        lo = lo < 0 ? 0 : lo;
        // the forward clamps every position beyond the last voxel onto it (arange_over_f grid with Xo > X * zoom): all of
        // those outputs read index maxi with weight 1 and must be gathered here, or this is not the adjoint
        hi = (hi > no - 1 || j == maxi) ? no - 1 : hi;
    } else {  // single input sample along this axis: every output reads index 0
        lo = 0;
        hi = no - 1;
    }
}
__device__ __forceinline__ float axis_weight(int i, float st, int maxi, int j)
{
    const AxisG a = axis_setup_g((float)i * st, maxi);
    return (a.i0 == j ? a.w0 : 0.f) + (a.i1 == j ? a.w1 : 0.f);
}

__global__ void __launch_bounds__(TB)
resize_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int X, int Y, int Z, int C, int Xo,
                  int Yo, int Zo, float stx, float sty, float stz, float mul)
{
    const int64_t nvo = (int64_t)Xo * Yo * Zo, nvi = (int64_t)X * Y * Z;
    const int64_t total = (int64_t)B * nvi * C;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TB) {
        const int c = (int)(i % C);
        const int64_t v = i / C;
        const int64_t b = v / nvi;
        const int64_t r = v - b * nvi;
        const int z = (int)(r % Z), y = (int)((r / Z) % Y), x = (int)(r / ((int64_t)Z * Y));
        int x0, x1, y0, y1, z0, z1;
        axis_range(x, stx, Xo, X - 1, x0, x1);
        axis_range(y, sty, Yo, Y - 1, y0, y1);
        axis_range(z, stz, Zo, Z - 1, z0, z1);
        const float* src = dout + b * nvo * C + c;
        float acc = 0.f;
        constexpr int RW = 10;   // candidates per axis for factors down to 1/2 (8); longer ranges take the generic loop
        if (z1 - z0 < RW) {
            // the z weights do not depend on (ix, iy): once per thread, in registers (they were re-derived per row)
            float wzv[RW];
#pragma unroll
            for (int k = 0; k < RW; ++k) wzv[k] = (z0 + k <= z1) ? axis_weight(z0 + k, stz, Z - 1, z) : 0.f;
            for (int ix = x0; ix <= x1; ++ix) {
                const float wx = axis_weight(ix, stx, X - 1, x);
                if (wx == 0.f) continue;
                for (int iy = y0; iy <= y1; ++iy) {
                    const float wy = axis_weight(iy, sty, Y - 1, y);
                    if (wy == 0.f) continue;
                    const float wxy = wx * wy;
                    const float* row = src + (((int64_t)ix * Yo + iy) * Zo + z0) * C;
#pragma unroll
                    for (int k = 0; k < RW; ++k)
                        if (wzv[k] != 0.f) acc += (wxy * wzv[k]) * (row[(int64_t)k * C] * mul);
                }
            }
        } else {
            for (int ix = x0; ix <= x1; ++ix) {
                const float wx = axis_weight(ix, stx, X - 1, x);
                if (wx == 0.f) continue;
                for (int iy = y0; iy <= y1; ++iy) {
                    const float wy = axis_weight(iy, sty, Y - 1, y);
                    if (wy == 0.f) continue;
                    const float wxy = wx * wy;
                    const float* row = src + ((int64_t)ix * Yo + iy) * Zo * C;
                    for (int iz = z0; iz <= z1; ++iz) {
                        const float wz = axis_weight(iz, stz, Z - 1, z);
                        if (wz != 0.f) acc += (wxy * wz) * (row[(int64_t)iz * C] * mul);
                    }
                }
            }
        }
        din[i] = acc;
    }
}

// The resize is separable (the weight of output i on input j is a product of one weight per axis), so its adjoint is the three
// per-axis adjoints one after the other: 2 .. 6 candidates per element and pass instead of their product (125 .. 216 gathers per
// element for the x2 resize in resize_bwd_kernel: 163 us for the 49 MB field gradient of a 160^3 step).  View of a pass:
// [outer][n][inner] with the contracted axis in the middle; every thread owns one element of the result and reads its candidates
// `inner` elements apart (coalesced along inner).  Same weights (axis_weight), fixed order, no atomics.
__global__ void __launch_bounds__(TB)
resize_bwd_axis_kernel(const float* __restrict__ dout, float* __restrict__ din, int64_t outer, int n_in, int n_out, int64_t inner,
                       float st, float mul)
{
    const int64_t total = outer * n_in * inner;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TB) {
        const int64_t in = i % inner;
        const int j = (int)((i / inner) % n_in);
        const int64_t o = i / (inner * n_in);
        int lo, hi;
        axis_range(j, st, n_out, n_in - 1, lo, hi);
        const float* src = dout + (o * n_out) * inner + in;
        float acc = 0.f;
        for (int k = lo; k <= hi; ++k) {
            const float w = axis_weight(k, st, n_in - 1, j);
            if (w != 0.f) acc += w * src[(int64_t)k * inner];
        }
        din[i] = acc * mul;
    }
}

// The same pass for inner >= 256: one (outer, j) line per workgroup row (blockIdx.y = j, blockIdx.z = outer index), threads along
// `inner`.  The candidate range and the (at most RBW) weights are those of the whole line: computed once per thread, then the
// element loop is loads and multiply-adds only -- no 64-bit index divisions and no floor / clamp per candidate and element.
constexpr int RBW = 8;
__global__ void __launch_bounds__(TB)
resize_bwd_axis_lines_kernel(const float* __restrict__ dout, float* __restrict__ din, int n_in, int n_out, int64_t inner, float st,
                             float mul)
{
    const int j = blockIdx.y;
    const int64_t o = blockIdx.z;
    int lo, hi;
    axis_range(j, st, n_out, n_in - 1, lo, hi);
    float w[RBW];
#pragma unroll
    for (int k = 0; k < RBW; ++k) w[k] = (lo + k <= hi) ? axis_weight(lo + k, st, n_in - 1, j) : 0.f;
    const float* src = dout + (o * n_out + lo) * inner;
    float* dst = din + (o * n_in + j) * inner;
    for (int64_t in = (int64_t)blockIdx.x * TB + threadIdx.x; in < inner; in += (int64_t)gridDim.x * TB) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < RBW; ++k)
            if (w[k] != 0.f) acc += w[k] * src[(int64_t)k * inner + in];   // w[k] == 0 beyond hi: no read past the tensor
        dst[in] = acc * mul;
    }
}

// host side of one pass: the line kernel where it applies (long inner, few candidates per line, grid dims in range)
static void launch_resize_bwd_axis(const float* dout, float* din, int64_t outer, int n_in, int n_out, int64_t inner, float st,
                                   float mul, hipStream_t stream)
{
    // candidates per line: ceil((j + 1) / st) - floor((j - 1) / st) + 1 <= 2 / st + 3 (st > 0), the whole axis when st == 0
    // st < 1 (the adjoint of an up-sampling: 5 - 6 candidates per line) is where the hoisted weights pay: 76 -> 63 us for the x2
    // resize of a 160^3 field; with one or two candidates per line (st = 2) the element-indexed kernel is the faster one (101 vs 106)
    bool few = st > 0.f && st < 1.f && 2.0f / st + 3.0f <= (float)RBW;
    if (few) {   // the last input index also gathers every output clamped onto it (axis_range: hi = n_out - 1 for j == n_in - 1)
        const int lo_last = (int)floorf((float)(n_in - 2) / st);
        few = n_out - (lo_last < 0 ? 0 : lo_last) <= RBW;
    }
    if (inner >= 256 && few && outer <= 65535 && n_in <= 65535) {
        int gx = (int)((inner + TB - 1) / TB);
        if (gx > 64) gx = 64;
        hipLaunchKernelGGL(resize_bwd_axis_lines_kernel, dim3(gx, n_in, (unsigned)outer), dim3(TB), 0, stream, dout, din, n_in, n_out,
                           inner, st, mul);
    } else {
        hipLaunchKernelGGL(resize_bwd_axis_kernel, dim3(stream_grid(outer * n_in * inner, TB)), dim3(TB), 0, stream, dout, din, outer,
                           n_in, n_out, inner, st, mul);
    }
}

// ------------------------------------------------------------------------- //
// adjoint of out = s*b + (s*a) o (id + s*b)   (3-channel fields)            //
//   da += s * scatter(w * dout);  db += s * (dout + sum_c dout_c * d interp(a_c)/d loc)
// For a == b (scaling and squaring) pass da == db.  da/db must be pre-zeroed. //
// ------------------------------------------------------------------------- //
__global__ void __launch_bounds__(TB)
compose_bwd_kernel(const float* __restrict__ a, const float* __restrict__ bf, const float* __restrict__ dout,
                   float* __restrict__ da, float* __restrict__ db, int B, int X, int Y, int Z, float s)
{
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t total = (int64_t)B * nvox;
    const int64_t sy = (int64_t)Z * 3, sx = (int64_t)Y * Z * 3;
    for (int64_t v = (int64_t)blockIdx.x * TB + threadIdx.x; v < total; v += (int64_t)gridDim.x * TB) {
        const int64_t b = v / nvox;
        const int64_t r = v - b * nvox;
        const int z = (int)(r % Z), y = (int)((r / Z) % Y), x = (int)(r / ((int64_t)Z * Y));
        const float* f = bf + v * 3;
        const AxisG ax = axis_setup_g((float)x + f[0] * s, X - 1);
        const AxisG ay = axis_setup_g((float)y + f[1] * s, Y - 1);
        const AxisG az = axis_setup_g((float)z + f[2] * s, Z - 1);
        const float g[3] = {dout[v * 3], dout[v * 3 + 1], dout[v * 3 + 2]};
        const float* abase = a + b * nvox * 3;
        float* dabase = da + b * nvox * 3;
        float dl[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int qx = cc >> 2, qy = (cc >> 1) & 1, qz = cc & 1;
            const int64_t off = (qx ? ax.i1 : ax.i0) * sx + (qy ? ay.i1 : ay.i0) * sy + (int64_t)(qz ? az.i1 : az.i0) * 3;
            const float wx = qx ? ax.w1 : ax.w0, wy = qy ? ay.w1 : ay.w0, wz = qz ? az.w1 : az.w0;
            const float w = (wx * wy) * wz;
            float dotv = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dotv += g[c] * (abase[off + c] * s);
                if (w != 0.f) atomicAdd(dabase + off + c, s * w * g[c]);
            }
            dl[0] += (qx ? 1.f : -1.f) * wy * wz * dotv;
            dl[1] += (qy ? 1.f : -1.f) * wx * wz * dotv;
            dl[2] += (qz ? 1.f : -1.f) * wx * wy * dotv;
        }
        atomicAdd(db + v * 3, s * (g[0] + ax.inr * dl[0]));
        atomicAdd(db + v * 3 + 1, s * (g[1] + ay.inr * dl[1]));
        atomicAdd(db + v * 3 + 2, s * (g[2] + az.inr * dl[2]));
    }
}

// Same gradients, scatter accumulated per tile in LDS: a workgroup owns an 8 x 8 x 16 voxel tile, the trilinear splat of
// d out / d a lands in an LDS image of the tile + 2 voxels of margin (LDS atomics), which is flushed to HBM with one atomic
// per touched entry instead of one per (voxel, corner): displacements of scaling-and-squaring steps are a few voxels at
// most, so almost every splat stays inside the image; anything farther goes straight to HBM as before.  db (the
// gradient through the sampling location and the additive term) is exclusive per voxel: it joins the LDS image when it
// aliases da (VecInt: out = v + warp(v, v)), else it is stored.
constexpr int CB_TX = 8, CB_TY = 8, CB_TZ = 16, CB_M = 2;
constexpr int CB_RX = CB_TX + 2 * CB_M, CB_RY = CB_TY + 2 * CB_M, CB_RZ = CB_TZ + 2 * CB_M;
constexpr int CB_REGION = CB_RX * CB_RY * CB_RZ;   // 2880 voxels x 3 floats = 34,560 B
__global__ void __launch_bounds__(TB)
compose_bwd_tiled_kernel(const float* __restrict__ a, const float* __restrict__ bf, const float* __restrict__ dout,
                         float* __restrict__ da, float* __restrict__ db, int B, int X, int Y, int Z, float s, int ntx,
                         int nty, int ntz)
{
    __shared__ float img[CB_REGION * 3];
    for (int i = threadIdx.x; i < CB_REGION * 3; i += TB) img[i] = 0.f;
    int t = blockIdx.x;
    const int tzi = t % ntz; t /= ntz;
    const int tyi = t % nty; t /= nty;
    const int txi = t % ntx;
    const int b = t / ntx;
    const int x0 = txi * CB_TX, y0 = tyi * CB_TY, z0 = tzi * CB_TZ;
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t sy = (int64_t)Z * 3, sx = (int64_t)Y * Z * 3;
    const float* abase = a + b * nvox * 3;
    float* dabase = da + b * nvox * 3;
    const bool alias = (da == db);
    __syncthreads();
    // The trilinear splat is LDS float atomics (~117 LDS cycles per ds_add_f32 wave-instruction): a 16-lane DPP row is one
    // z line of the tile, and where the field is smooth the UPPER z corner of voxel z is the LOWER z corner of voxel z + 1 -- then
    // lane z + 1 adds both values with one atomic and lane z issues none for that corner: 12 - 15 atomics per voxel instead of 24.
    auto shr1f = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true)); };
    auto shr1i = [](int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); };   // row_shr:1, 0 into the row's first lane
    auto shl1i = [](int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, true); };   // row_shl:1, 0 into its last lane
    static_assert(CB_TZ == 16 && (CB_TX * CB_TY * CB_TZ) % TB == 0, "a DPP row = one z line of the tile; every lane runs every step");
    for (int lv = threadIdx.x; lv < CB_TX * CB_TY * CB_TZ; lv += TB) {
        const int lz = lv % CB_TZ, ly = (lv / CB_TZ) % CB_TY, lx = lv / (CB_TZ * CB_TY);
        const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
        const bool active = x < X && y < Y && z < Z;
        const int cxv = active ? x : 0, cyv = active ? y : 0, czv = active ? z : 0;      // inactive lanes: any valid voxel, values zeroed
        const int64_t v = b * nvox + ((int64_t)cxv * Y + cyv) * Z + czv;
        const float* f = bf + v * 3;
        const AxisG ax = axis_setup_g((float)cxv + f[0] * s, X - 1);
        const AxisG ay = axis_setup_g((float)cyv + f[1] * s, Y - 1);
        const AxisG az = axis_setup_g((float)czv + f[2] * s, Z - 1);
        const float g[3] = {active ? dout[v * 3] : 0.f, active ? dout[v * 3 + 1] : 0.f, active ? dout[v * 3 + 2] : 0.f};
        float dl[3] = {0.f, 0.f, 0.f};
        // voxel offsets fit 31 bits here (offsets of a 3-channel field of fewer than 2^31 / 3 voxels) -- else no merging
        const bool small = (int64_t)B * nvox * 3 < (int64_t)0x7ffffff0;
        const int lane_ = threadIdx.x & 63, row4 = lane_ >> 4;       // the wave's four z lines are four consecutive y
        const int up_addr = ((lane_ - 16) & 63) * 4, dn_addr = ((lane_ + 16) & 63) * 4;
#pragma unroll
        for (int qx = 0; qx < 2; ++qx) {
            const int ix = qx ? ax.i1 : ax.i0;
            const float wx = qx ? ax.w1 : ax.w0;
            const int rx = ix - x0 + CB_M;
            // pending adds of this lane: [qy][qz] -> key (voxel offset + 1), three values, target, "still to be issued by me"
            int key[2][2], ro[2][2];
            int64_t off[2][2];
            float val[2][2][3];
            bool pend[2][2], inq[2][2], mine[2][2];
#pragma unroll
            for (int qy = 0; qy < 2; ++qy) {
                const int iy = qy ? ay.i1 : ay.i0;
                const float wy = qy ? ay.w1 : ay.w0;
                const int64_t offxy = ix * sx + iy * sy;
                const int ry = iy - y0 + CB_M;
                const bool inxy = (unsigned)rx < (unsigned)CB_RX && (unsigned)ry < (unsigned)CB_RY;
                float wq[2];
#pragma unroll
                for (int qz = 0; qz < 2; ++qz) {
                    const int iz = qz ? az.i1 : az.i0;
                    const float wz = qz ? az.w1 : az.w0;
                    off[qy][qz] = offxy + (int64_t)iz * 3;
                    wq[qz] = (wx * wy) * wz;
                    const int rz = iz - z0 + CB_M;
                    inq[qy][qz] = inxy && (unsigned)rz < (unsigned)CB_RZ;
                    ro[qy][qz] = ((rx * CB_RY + ry) * CB_RZ + rz) * 3;
                    key[qy][qz] = (int)(b * nvox * 3 + off[qy][qz]) + 1;
                    float dotv = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) dotv += g[c] * (abase[off[qy][qz] + c] * s);
                    dl[0] += (qx ? 1.f : -1.f) * wy * wz * dotv;
                    dl[1] += (qy ? 1.f : -1.f) * wx * wz * dotv;
                    dl[2] += (qz ? 1.f : -1.f) * wx * wy * dotv;
                }
                // z: the lower lane's upper corner is my lower corner -> I add both, it adds nothing there
                const bool take = active && small && shr1i((active && small) ? key[qy][1] : 0) == key[qy][0];
                const bool given = shl1i(take ? 1 : 0) != 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v1 = s * wq[1] * g[c];
                    const float pv = shr1f(v1);
                    val[qy][0][c] = s * wq[0] * g[c] + (take ? pv : 0.f);
                    val[qy][1][c] = v1;
                }
                pend[qy][0] = active && (wq[0] != 0.f || take);
                pend[qy][1] = active && wq[1] != 0.f && !given;
                mine[qy][0] = active;
                mine[qy][1] = active && !given;
            }
            // y: the same between the wave's z lines -- the y + 1 line takes what the y line has pending for the same voxel
#pragma unroll
            for (int qz = 0; qz < 2; ++qz) {
                const int hand = (pend[1][qz] && small) ? key[1][qz] : 0;
                const int upkey = __builtin_amdgcn_ds_bpermute(up_addr, hand);
                const bool takey = row4 > 0 && mine[0][qz] && upkey != 0 && upkey == key[0][qz];
                const bool giveny = __builtin_amdgcn_ds_bpermute(dn_addr, takey ? 1 : 0) != 0 && row4 < 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float upv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(up_addr, __builtin_bit_cast(int, val[1][qz][c])));
                    if (takey) val[0][qz][c] += upv;
                }
                pend[0][qz] = pend[0][qz] || takey;
                pend[1][qz] = pend[1][qz] && !giveny;
            }
#pragma unroll
            for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                for (int qz = 0; qz < 2; ++qz)
                    if (pend[qy][qz]) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            if (inq[qy][qz]) atomicAdd(&img[ro[qy][qz] + c], val[qy][qz][c]);
                            else atomicAdd(dabase + off[qy][qz] + c, val[qy][qz][c]);
                        }
                    }
        }
        if (!active) continue;
        const float d0 = s * (g[0] + ax.inr * dl[0]), d1 = s * (g[1] + ay.inr * dl[1]), d2 = s * (g[2] + az.inr * dl[2]);
        if (alias) {
            float* o = img + (((lx + CB_M) * CB_RY + ly + CB_M) * CB_RZ + lz + CB_M) * 3;
            atomicAdd(o, d0); atomicAdd(o + 1, d1); atomicAdd(o + 2, d2);
        } else {
            db[v * 3] = d0; db[v * 3 + 1] = d1; db[v * 3 + 2] = d2;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CB_REGION * 3; i += TB) {
        const float val = img[i];
        if (val == 0.f) continue;
        const int c = i % 3, r = i / 3;
        const int rz = r % CB_RZ, ry = (r / CB_RZ) % CB_RY, rx = r / (CB_RZ * CB_RY);
        const int gx = x0 + rx - CB_M, gy = y0 + ry - CB_M, gz = z0 + rz - CB_M;
        if (gx < 0 || gx >= X || gy < 0 || gy >= Y || gz < 0 || gz >= Z) continue;   // never touched: indices are clamped
        atomicAdd(dabase + (int64_t)gx * sx + (int64_t)gy * sy + (int64_t)gz * 3 + c, val);
    }
}

// general linear-warp gradient w.r.t. the flow: dflow[b,v,:] = sum_c dout[b,v,c] * d interp(vol_c)/d loc
__global__ void __launch_bounds__(TB)
warp_bwd_flow_kernel(const float* __restrict__ vol, const float* __restrict__ flow, const float* __restrict__ dout,
                     float* __restrict__ dflow, int B, int X, int Y, int Z, int C)
{
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t total = (int64_t)B * nvox;
    const int64_t sz = C, sy = (int64_t)Z * C, sx = (int64_t)Y * Z * C;
    for (int64_t v = (int64_t)blockIdx.x * TB + threadIdx.x; v < total; v += (int64_t)gridDim.x * TB) {
        const int64_t b = v / nvox;
        const int64_t r = v - b * nvox;
        const int z = (int)(r % Z), y = (int)((r / Z) % Y), x = (int)(r / ((int64_t)Z * Y));
        const AxisG ax = axis_setup_g((float)x + flow[v * 3], X - 1);
        const AxisG ay = axis_setup_g((float)y + flow[v * 3 + 1], Y - 1);
        const AxisG az = axis_setup_g((float)z + flow[v * 3 + 2], Z - 1);
        const float* base = vol + b * nvox * C;
        float dl[3] = {0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) {
            const float g = dout[v * C + c];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const int qx = cc >> 2, qy = (cc >> 1) & 1, qz = cc & 1;
                const float val = g * base[(qx ? ax.i1 : ax.i0) * sx + (qy ? ay.i1 : ay.i0) * sy + (qz ? az.i1 : az.i0) * sz + c];
                const float wx = qx ? ax.w1 : ax.w0, wy = qy ? ay.w1 : ay.w0, wz = qz ? az.w1 : az.w0;
                dl[0] += (qx ? 1.f : -1.f) * wy * wz * val;
                dl[1] += (qy ? 1.f : -1.f) * wx * wz * val;
                dl[2] += (qz ? 1.f : -1.f) * wx * wy * val;
            }
        }
        dflow[v * 3] = ax.inr * dl[0];
        dflow[v * 3 + 1] = ay.inr * dl[1];
        dflow[v * 3 + 2] = az.inr * dl[2];
    }
}

// linear-warp gradient w.r.t. the volume (scatter-add); dvol pre-zeroed
__global__ void __launch_bounds__(TB)
warp_bwd_vol_kernel(const float* __restrict__ flow, const float* __restrict__ dout, float* __restrict__ dvol, int B,
                    int X, int Y, int Z, int C)
{
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t total = (int64_t)B * nvox * C;
    const int64_t sz = C, sy = (int64_t)Z * C, sx = (int64_t)Y * Z * C;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TB) {
        const int c = (int)(i % C);
        const int64_t v = i / C;
        const int64_t b = v / nvox;
        const int64_t r = v - b * nvox;
        const int z = (int)(r % Z), y = (int)((r / Z) % Y), x = (int)(r / ((int64_t)Z * Y));
        const AxisG ax = axis_setup_g((float)x + flow[v * 3], X - 1);
        const AxisG ay = axis_setup_g((float)y + flow[v * 3 + 1], Y - 1);
        const AxisG az = axis_setup_g((float)z + flow[v * 3 + 2], Z - 1);
        const float g = dout[i];
        float* base = dvol + b * nvox * C + c;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const int qx = cc >> 2, qy = (cc >> 1) & 1, qz = cc & 1;
            const float w = ((qx ? ax.w1 : ax.w0) * (qy ? ay.w1 : ay.w0)) * (qz ? az.w1 : az.w0);
            if (w != 0.f)
                atomicAdd(base + (qx ? ax.i1 : ax.i0) * sx + (qy ? ay.i1 : ay.i0) * sy + (qz ? az.i1 : az.i0) * sz, w * g);
        }
    }
}

// ------------------------------------------------------------------------- //
// conv plumbing                                                             //
// ------------------------------------------------------------------------- //
// dz = dy * (y >= 0 ? 1 : alpha) (in place allowed), db partial sums per block
__global__ void __launch_bounds__(TB)
leaky_bwd_bias_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dz,
                      double* __restrict__ part, int64_t nvox, int C, float alpha, int leaky, int nblk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s = reinterpret_cast<double*>(smem);  // [TB]
    const int T = (TB / C) * C;  // C <= 256; threads >= T idle so that a thread's channel is fixed
    const int64_t nel = nvox * C;
    const int64_t chunk = (((nel + nblk - 1) / nblk + T - 1) / T) * T;
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    const int64_t hi = lo + chunk < nel ? lo + chunk : nel;
    // double from the first addition on: the flow head's bias gradient is this sum over every voxel with 40 - 80x cancellation
    double acc = 0.0;
    if ((int)threadIdx.x < T) {
        for (int64_t e = lo + threadIdx.x; e < hi; e += T) {
            float g = dy[e];
            if (leaky && y[e] < 0.f) g *= alpha;
            dz[e] = g;
            acc += (double)g;
        }
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        double r = 0.0;
        for (int k = threadIdx.x; k < T; k += C) r += s[k];
        part[(int64_t)blockIdx.x * C + threadIdx.x] = r;
    }
}

// one wave per channel: ordered strided partial sums, then a wave reduction
__global__ void __launch_bounds__(64)
bias_final_kernel(const double* __restrict__ part, float* __restrict__ db, int C, int nblk, int accumulate)
{
    const int c = blockIdx.x;
    double r = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) r += part[(int64_t)k * C + c];
    r = wave_sum(r);
    if (threadIdx.x == 0) {
        if (accumulate) db[c] += (float)r; else db[c] = (float)r;
    }
}

// split the gradient of concat([up2(in0), in1]): d_in0 = 2x2x2 sum-pool of dcat[..., :C0] ; d_in1 (+)= dcat[..., C0:]
__global__ void __launch_bounds__(TB)
upcat_bwd_kernel(const float* __restrict__ dcat, float* __restrict__ d0, float* __restrict__ d1, int B, int X, int Y,
                 int Z, int C0, int C1, int up0, int acc1)
{
    const int C = C0 + C1;
    const int64_t nvox = (int64_t)B * X * Y * Z;
    // part 1: skip channels (and in0 when not upsampled)
    const int64_t tot1 = nvox * C;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < tot1; i += (int64_t)gridDim.x * TB) {
        const int c = (int)(i % C);
        const int64_t v = i / C;
        if (c >= C0) {
            float* o = d1 + v * C1 + (c - C0);
            if (acc1) *o += dcat[i]; else *o = dcat[i];
        } else if (!up0) {
            d0[v * C0 + c] = dcat[i];
        }
    }
    if (!up0) return;
    const int Xh = X / 2, Yh = Y / 2, Zh = Z / 2;
    const int64_t tot0 = (int64_t)B * Xh * Yh * Zh * C0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < tot0; i += (int64_t)gridDim.x * TB) {
        int64_t r = i;
        const int c = (int)(r % C0); r /= C0;
        const int z = (int)(r % Zh); r /= Zh;
        const int y = (int)(r % Yh); r /= Yh;
        const int x = (int)(r % Xh);
        const int b = (int)(r / Xh);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int gx = 2 * x + (k >> 2), gy = 2 * y + ((k >> 1) & 1), gz = 2 * z + (k & 1);
            s += dcat[((((int64_t)b * X + gx) * Y + gy) * Z + gz) * C + c];
        }
        d0[i] = s;
    }
}

// dx (+)= route(dpool) to the first maximum of each 2x2x2 window of x (scan order x,y,z)
__global__ void __launch_bounds__(TB)
maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dpool, float* __restrict__ dx, int B, int X,
                   int Y, int Z, int C, int accumulate)
{
    const int Xo = X / 2, Yo = Y / 2, Zo = Z / 2;
    const int64_t total = (int64_t)B * Xo * Yo * Zo * C;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TB) {
        int64_t r = i;
        const int c = (int)(r % C); r /= C;
        const int z = (int)(r % Zo); r /= Zo;
        const int y = (int)(r % Yo); r /= Yo;
        const int xx = (int)(r % Xo);
        const int b = (int)(r / Xo);
        float best = -INFINITY;
        int bk = 0;
        int64_t offs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int gx = 2 * xx + (k >> 2), gy = 2 * y + ((k >> 1) & 1), gz = 2 * z + (k & 1);
            offs[k] = ((((int64_t)b * X + gx) * Y + gy) * Z + gz) * C + c;
            const float v = x[offs[k]];
            if (v > best) { best = v; bk = k; }
        }
        const float g = dpool[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float val = (k == bk) ? g : 0.f;
            if (accumulate) dx[offs[k]] += val; else dx[offs[k]] = val;
        }
    }
}

// ---- float4 versions with the LeakyReLU backward of the receiving layers fused in ------------------------- //
// A gradient contribution c to an activated tensor y reaches the layer's pre-activation as c * L'(y) and its bias
// as sum_v c * L'(y): both are linear in c, so every producer of a contribution can apply them itself and the
// separate leaky_bwd_bias pass disappears.  y pointers may be null (plain split / routing).
__device__ __forceinline__ float4 mask4(float4 g, float4 y, float alpha)
{
    if (y.x < 0.f) g.x *= alpha;
    if (y.y < 0.f) g.y *= alpha;
    if (y.z < 0.f) g.z *= alpha;
    if (y.w < 0.f) g.w *= alpha;
    return g;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// block-level column sums: thread t < T owns channel quad t % Q; part[blockIdx.x][4 Q] in double
__device__ __forceinline__ void quad_colsum(float4 cs, float4* s4, int Q, int T, double* __restrict__ part)
{
    __syncthreads();
    s4[threadIdx.x] = cs;
    __syncthreads();
    if ((int)threadIdx.x < Q) {
        double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
        for (int k = threadIdx.x; k < T; k += Q) {
            const float4 v = s4[k];
            r0 += (double)v.x; r1 += (double)v.y; r2 += (double)v.z; r3 += (double)v.w;
        }
        double* o = part + (int64_t)blockIdx.x * (4 * Q) + 4 * threadIdx.x;
        o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
    }
}

__global__ void __launch_bounds__(TB)
upcat_bwd_v4_kernel(const float* __restrict__ dcat, float* __restrict__ d0, float* __restrict__ d1, int B, int X, int Y,
                    int Z, int C0, int C1, int up0, int acc1, const float* __restrict__ y0, const float* __restrict__ y1,
                    float alpha, double* __restrict__ part0, double* __restrict__ part1)
{
    __shared__ float4 s4[TB];
    const int C = C0 + C1;
    const int64_t nvox = (int64_t)B * X * Y * Z;
    const int tid = threadIdx.x;
    if (C1 > 0) {  // skip channels: d1 (+)= dcat[..., C0:] * L'(y1)
        const int Q = C1 >> 2, T = (TB / Q) * Q, vpb = T / Q;
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < T) {
            const int q = tid % Q;
            for (int64_t v = (int64_t)blockIdx.x * vpb + tid / Q; v < nvox; v += (int64_t)gridDim.x * vpb) {
                float4 g = *reinterpret_cast<const float4*>(dcat + v * C + C0 + 4 * q);
                if (y1) {
                    g = mask4(g, *reinterpret_cast<const float4*>(y1 + v * C1 + 4 * q), alpha);
                    cs = add4(cs, g);
                }
                float4* o = reinterpret_cast<float4*>(d1 + v * C1 + 4 * q);
                *o = acc1 ? add4(*o, g) : g;
            }
        }
        if (y1) quad_colsum(cs, s4, Q, T, part1);
    }
    const int Q = C0 >> 2, T = (TB / Q) * Q, vpb = T / Q;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!up0) {
        if (tid < T) {
            const int q = tid % Q;
            for (int64_t v = (int64_t)blockIdx.x * vpb + tid / Q; v < nvox; v += (int64_t)gridDim.x * vpb) {
                float4 g = *reinterpret_cast<const float4*>(dcat + v * C + 4 * q);
                if (y0) {
                    g = mask4(g, *reinterpret_cast<const float4*>(y0 + v * C0 + 4 * q), alpha);
                    cs = add4(cs, g);
                }
                *reinterpret_cast<float4*>(d0 + v * C0 + 4 * q) = g;
            }
        }
    } else {
        const int Xh = X / 2, Yh = Y / 2, Zh = Z / 2;
        const int64_t nh = (int64_t)B * Xh * Yh * Zh;
        if (tid < T) {
            const int q = tid % Q;
            for (int64_t v = (int64_t)blockIdx.x * vpb + tid / Q; v < nh; v += (int64_t)gridDim.x * vpb) {
                int64_t r = v;
                const int z = (int)(r % Zh); r /= Zh;
                const int y = (int)(r % Yh); r /= Yh;
                const int x = (int)(r % Xh);
                const int b = (int)(r / Xh);
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int gx = 2 * x + (k >> 2), gy = 2 * y + ((k >> 1) & 1), gz = 2 * z + (k & 1);
                    g = add4(g, *reinterpret_cast<const float4*>(dcat + ((((int64_t)b * X + gx) * Y + gy) * Z + gz) * C + 4 * q));
                }
                if (y0) {
                    g = mask4(g, *reinterpret_cast<const float4*>(y0 + v * C0 + 4 * q), alpha);
                    cs = add4(cs, g);
                }
                *reinterpret_cast<float4*>(d0 + v * C0 + 4 * q) = g;
            }
        }
    }
    if (y0) quad_colsum(cs, s4, Q, T, part0);
}

// dx (+)= route(dpool) to the first maximum of each 2x2x2 window, times L'(x at that maximum) when `masked`
// (x is then the activated output whose gradient dx is)
__global__ void __launch_bounds__(TB)
maxpool_bwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ dpool, float* __restrict__ dx, int B, int X,
                      int Y, int Z, int C, int accumulate, int masked, float alpha, double* __restrict__ part)
{
    __shared__ float4 s4[TB];
    const int Xo = X / 2, Yo = Y / 2, Zo = Z / 2;
    const int64_t nh = (int64_t)B * Xo * Yo * Zo;
    const int Q = C >> 2, T = (TB / Q) * Q, vpb = T / Q;
    const int tid = threadIdx.x;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < T) {
        const int q = tid % Q;
        for (int64_t v = (int64_t)blockIdx.x * vpb + tid / Q; v < nh; v += (int64_t)gridDim.x * vpb) {
            int64_t r = v;
            const int z = (int)(r % Zo); r /= Zo;
            const int y = (int)(r % Yo); r /= Yo;
            const int xx = (int)(r % Xo);
            const int b = (int)(r / Xo);
            float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            int k0 = 0, k1 = 0, k2 = 0, k3 = 0;
            int64_t offs[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int gx = 2 * xx + (k >> 2), gy = 2 * y + ((k >> 1) & 1), gz = 2 * z + (k & 1);
                offs[k] = ((((int64_t)b * X + gx) * Y + gy) * Z + gz) * C + 4 * q;
                const float4 val = *reinterpret_cast<const float4*>(x + offs[k]);
                if (val.x > best.x) { best.x = val.x; k0 = k; }
                if (val.y > best.y) { best.y = val.y; k1 = k; }
                if (val.z > best.z) { best.z = val.z; k2 = k; }
                if (val.w > best.w) { best.w = val.w; k3 = k; }
            }
            float4 g = *reinterpret_cast<const float4*>(dpool + v * C + 4 * q);
            if (masked) {
                g = mask4(g, best, alpha);
                cs = add4(cs, g);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float4 val = make_float4(k == k0 ? g.x : 0.f, k == k1 ? g.y : 0.f, k == k2 ? g.z : 0.f, k == k3 ? g.w : 0.f);
                float4* o = reinterpret_cast<float4*>(dx + offs[k]);
                *o = accumulate ? add4(*o, val) : val;
            }
        }
    }
    if (masked) quad_colsum(cs, s4, Q, T, part);
}

// ------------------------------------------------------------------------- //
// wgrad on the matrix cores: dW[tap][ci][co] = sum_v Xpad[v+tap][ci] * dZ[v][co]
// ------------------------------------------------------------------------- //
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int W_TX = 4, W_TY = 8, W_TZ = 8;
constexpr int W_HX = 6, W_HY = 10, W_HZ = 10;
constexpr int W_HROWS = W_HX * W_HY * W_HZ;
constexpr int W_ROWB = 144;          // 32 fp32 channels + 16 B pad
constexpr int W_A_BYTES = W_HROWS * W_ROWB;
constexpr int W_DZROW = 272;         // 64 fp32 couts + 16 B pad
constexpr int W_B_BYTES = 256 * W_DZROW;
constexpr int W_THREADS = 512;
constexpr int W_UNITS = 7;           // ceil(27 taps * 2 co-tiles / 8 waves)

struct WgradParams {
    const float* in0;
    const float* in1;
    const float* dz;
    float* slab;  // [gridDim.x][nslices][ncob][27][32][64]
    int B, X, Y, Z, C0, C1, up0, Cout;
    int ntx, nty, ntz, ntiles;
    unsigned bytes_src[2], bytes_dz;   // operand sizes for the buffer-descriptor path (0 = larger than 3.75 GB: plain loads)
};

// Wave w owns the accumulator tiles (tap = (w >> 1) + 4 j, co-tile = w & 1), j = 0..6 (units 54, 55 of the
// 8 x 7 grid are padding and never stored) -> per k-step (2 voxels): ONE dZ read, 7 X reads, 7 MFMAs, no
// branches.  The next voxel tile is prefetched into registers while the current one is consumed.
template <int COT>  // co-tiles of 32 per 64-wide block: 2 (general) or 1 (Cout <= 32, e.g. the flow head)
__global__ void __launch_bounds__(W_THREADS, 2)
wgrad_kernel(const WgradParams p)
{
    constexpr int NU = (COT == 2) ? 7 : 4;  // accumulator tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + W_A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.y, cob = blockIdx.z;
    const int ch0 = slice * 32;
    const bool first = ch0 < p.C0;
    const float* src = first ? p.in0 : p.in1;
    const int Cs = first ? p.C0 : p.C1;
    const int chs = first ? ch0 : ch0 - p.C0;
    const bool up = first && p.up0;
    const int X2 = p.X >> 1, Y2 = p.Y >> 1, Z2 = p.Z >> 1;
    constexpr int A_IT = (W_HROWS * 8 + W_THREADS - 1) / W_THREADS;  // 10
    constexpr int B_IT = 256 * 16 / W_THREADS;                       // 8

    f32x16 acc[NU];
    int tapoff[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        int tap = (COT == 2) ? (wave >> 1) + 4 * j : wave + 8 * j;
        if (tap > 26) tap = 26;  // padding unit: valid address, result discarded
        tapoff[j] = ((tap / 9) * (W_HY * W_HZ) + ((tap / 3) % 3) * W_HZ + (tap % 3)) * W_ROWB;
    }
    const int a_lane = (lane >> 5) * W_ROWB + (lane & 31) * 4;
    const int cot = (COT == 2) ? (wave & 1) : 0;
    const int b_lane = (lane >> 5) * W_DZROW + (cot * 32 + (lane & 31)) * 4;

    auto tile_origin = [&](int tile, int& b, int& x0, int& y0, int& z0) {
        int t = tile;
        const int tzi = t % p.ntz; t /= p.ntz;
        const int tyi = t % p.nty; t /= p.nty;
        const int txi = t % p.ntx;
        b = t / p.ntx;
        x0 = txi * W_TX; y0 = tyi * W_TY; z0 = tzi * W_TZ;
    };
    auto load_a = [&](int b, int x0, int y0, int z0, int it) -> float4 {
        const int i = tid + it * W_THREADS;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < W_HROWS * 8) {
            const int row = i >> 3, chunk = i & 7;
            const int hx = row / (W_HY * W_HZ), hy = (row / W_HZ) % W_HY, hz = row % W_HZ;
            const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
            if (gx >= 0 && gx < p.X && gy >= 0 && gy < p.Y && gz >= 0 && gz < p.Z) {
                size_t vox;
                if (up) vox = (((size_t)b * X2 + (gx >> 1)) * Y2 + (gy >> 1)) * Z2 + (gz >> 1);
                else vox = (((size_t)b * p.X + gx) * p.Y + gy) * p.Z + gz;
                val = *reinterpret_cast<const float4*>(src + vox * Cs + chs + chunk * 4);
            }
        }
        return val;
    };
    auto load_b = [&](int b, int x0, int y0, int z0, int it) -> float4 {
        const int i = tid + it * W_THREADS;
        const int v = i >> 4, chunk = i & 15;
        const int gx = x0 + (v >> 6), gy = y0 + ((v >> 3) & 7), gz = z0 + (v & 7);
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gx < p.X && gy < p.Y && gz < p.Z) {
            const float* q = p.dz + ((((size_t)b * p.X + gx) * p.Y + gy) * p.Z + gz) * p.Cout;
            const int co = cob * 64 + chunk * 4;
            if (co + 3 < p.Cout) val = *reinterpret_cast<const float4*>(q + co);
            else {
                if (co < p.Cout) val.x = q[co];
                if (co + 1 < p.Cout) val.y = q[co + 1];
                if (co + 2 < p.Cout) val.z = q[co + 2];
            }
        }
        return val;
    };
    auto store_a = [&](int it, float4 val) {
        const int i = tid + it * W_THREADS;
        if (i < W_HROWS * 8) *reinterpret_cast<float4*>(sA + (i >> 3) * W_ROWB + (i & 7) * 16) = val;
    };
    auto store_b = [&](int it, float4 val) {
        const int i = tid + it * W_THREADS;
        *reinterpret_cast<float4*>(sB + (i >> 4) * W_DZROW + (i & 15) * 16) = val;
    };

    float4 pa[A_IT], pb[B_IT];
    // passes of the persistent grid; every XCD walks a contiguous run of each pass's tiles (common.hpp: xcd_tile)
    int pass = 0, nxt = -1, tile = xcd_tile((int)blockIdx.x, (int)gridDim.x, 0, p.ntiles);
    if (tile >= 0) {
        int b, x0, y0, z0;
        tile_origin(tile, b, x0, y0, z0);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) pa[it] = load_a(b, x0, y0, z0, it);
#pragma unroll
        for (int it = 0; it < B_IT; ++it) pb[it] = load_b(b, x0, y0, z0, it);
    }
    for (; tile >= 0; tile = nxt, ++pass) {
        __syncthreads();  // everyone is done with the previous tile's LDS image
#pragma unroll
        for (int it = 0; it < A_IT; ++it) store_a(it, pa[it]);
#pragma unroll
        for (int it = 0; it < B_IT; ++it) store_b(it, pb[it]);
        __syncthreads();
        nxt = xcd_tile((int)blockIdx.x, (int)gridDim.x, pass + 1, p.ntiles);
        if (nxt >= 0) {  // register prefetch of the next tile, hidden under this tile's MFMAs
            int b, x0, y0, z0;
            tile_origin(nxt, b, x0, y0, z0);
#pragma unroll
            for (int it = 0; it < A_IT; ++it) pa[it] = load_a(b, x0, y0, z0, it);
#pragma unroll
            for (int it = 0; it < B_IT; ++it) pb[it] = load_b(b, x0, y0, z0, it);
        }
        for (int vx = 0; vx < W_TX; ++vx)
            for (int vy = 0; vy < W_TY; ++vy) {
                const int hbase = ((vx * W_HY + vy) * W_HZ) * W_ROWB + a_lane;
                const char* bp = sB + ((vx << 6) | (vy << 3)) * W_DZROW + b_lane;
                const char* ap[NU];
#pragma unroll
                for (int j = 0; j < NU; ++j) ap[j] = sA + hbase + tapoff[j];
#pragma unroll
                for (int vz = 0; vz < W_TZ; vz += 2) {
                    const float bv = *reinterpret_cast<const float*>(bp + vz * W_DZROW);
                    float av[NU];
#pragma unroll
                    for (int j = 0; j < NU; ++j) av[j] = *reinterpret_cast<const float*>(ap[j] + vz * W_ROWB);
#pragma unroll
                    for (int j = 0; j < NU; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv, acc[j], 0, 0, 0);
                }
            }
    }
    // slab[blk][slice][cob][tap][ci 32][co 64]
    float* out = p.slab + (((size_t)blockIdx.x * gridDim.y + slice) * gridDim.z + cob) * (27 * 32 * 64);
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const int tap = (COT == 2) ? (wave >> 1) + 4 * j : wave + 8 * j;
        if (tap < 27) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(tap * 32 + ci) * 64 + cot * 32 + (lane & 31)] = acc[j][r];
            }
        }
    }
}

// ---- fp32-grade wgrad on the bf16 matrix cores (fp32x3) --------------------------------------------- //
// Same decomposition as wgrad_kernel, but X and dZ are split into bf16 (hi, lo) while being staged and
// every product is three v_mfma_f32_32x32x16_bf16 (lo*hi + hi*lo + hi*hi).  Both MFMA operands need the
// voxel (k) index contiguous per lane while LDS holds [voxel][channel] rows, so fragments are fetched with
// the gfx950 transposing read ds_read_b64_tr_b16 (4 voxels x 16 channels per 16-lane group; semantics
// probed on hardware with tools/tr_probe.hip).  LDS images (conflict-free for the transposed reads):
//   sX row (128 B) = two 64-B planes (hi, lo), physical plane = plane ^ ((hz >> 1) & 1)
//   sZ row (256 B) = four 64-B segments (plane*2 + co/32), physical segment = segment ^ (vz & 3)
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

__device__ __forceinline__ s16x4 tr_read(const char* p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

__device__ __forceinline__ bf16x8_t frag8(s16x4 a, s16x4 b)
{
    s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ void split_bf16(float f, unsigned short& hi, unsigned short& lo)
{
    hi = f32_to_bf16(f);
    lo = f32_to_bf16(f - bf16_to_f32(hi));
}

constexpr int WX_A_BYTES = W_HROWS * 128;  // 76800
constexpr int WX_B_BYTES = 256 * 256;      // 65536

#ifdef MMR_DIAG
extern int g_diag_stamps;   // conv3d.hip
#endif
// Diagnostic (-DMMR_DIAG build + mmr_debug_set_stamps(1)): per wave slot the s_memtime sums over all workgroups of [0] wait at the top-of-tile
// barrier, [1] hi/lo split + LDS stores + barrier, [2] issue of the next tile's loads, [3] the 16 k-blocks, [4] tiles.
__device__ unsigned long long g_wgrad_stamp[8][8];
__device__ __forceinline__ unsigned long long wstamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// PF: which operand tiles of the NEXT voxel tile are prefetched into registers under the k-loop (bit 0: X, 40 VGPRs;
// bit 1: dZ, 32 VGPRs).  With both (72 VGPRs next to 112 accumulator registers) the kernel spills, and every spill
// reload between the prefetch loads is a `s_waitcnt vmcnt(0)`: the 13 loads of a tile ran as 13 dependent round trips
// (16 k of the 45 k cycles per tile, tools/wgrad_stamps.py).
// FULLCO: Cout is a multiple of 64, every float4 of dZ exists -> one branch-free load path.  (With the tail path compiled
// into the same kernel behind a wave-uniform branch, hipcc put `s_waitcnt vmcnt(0)` in front of every dZ load at the
// join of the two paths: 8 more dependent round trips per tile.)
template <int N> struct IC { static constexpr int value = N; };
// FOLD (round 3): the weight gradient of the UPSAMPLED half of a decoder layer, one parity class per workgroup.  With the
// folded weights Wf[p][s] (conv3d.hip, CV_UPFOLD) dWf[p][s] = sum_i x_low[i - 1 + p + s] (x) dz[2 i + p]: a correlation on the
// LOW-resolution grid between x_low (haloed tile, halo offset p + s per axis: one of the 27 ordinary tap offsets) and the class's
// sub-lattice of the full-resolution dz.  blockIdx.z = cob * 2 + class group; a workgroup takes FOUR classes per voxel tile (x_low
// staged once, then per class its dz tile and a k-loop); the 8 (p, s) pairs of a class are spread over the four wave pairs (two
// accumulator tiles per wave and class, eight in all instead of seven), p.X / Y / Z are the low-resolution dims, dz is addressed at
// 2 i + p of a [B, 2X, 2Y, 2Z, Cout] tensor, and the slab keeps the 27-tap layout (entry p + s).  wgrad_fold_reduce_kernel then
// adds every dWf[p][s] to the original taps it covers.  64 (class, tap) correlations over N / 8 voxels instead of 27 over N.
template <int COT, bool LO, bool STAMP = false, int PF = 3, bool FULLCO = false, bool BUF = false, bool FOLD = false>
__global__ void __launch_bounds__(W_THREADS, 2)
wgrad_x3_kernel(const WgradParams p)
{
    constexpr bool PFX = (PF & 1) != 0, PFZ = (PF & 2) != 0;
    unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st_t = 0;
    static_assert(!FOLD || (COT == 2 && FULLCO && BUF), "folded wgrad: 64-column blocks, buffer loads");
    constexpr int FNC = 4;                                  // FOLD: parity classes per workgroup (two accumulator tiles per wave each)
    constexpr int NU = FOLD ? 2 * FNC : (COT == 2) ? 7 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;
    char* sZ = smem + WX_A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.y, cob = FOLD ? (int)blockIdx.z >> 1 : (int)blockIdx.z;
    const int fcls0 = FOLD ? ((int)blockIdx.z & 1) * FNC : 0;   // FOLD: this workgroup's parity classes fcls0 .. fcls0 + 3
    // tap (27-index = halo offset) of accumulator tile j of this wave; FOLD: tile j belongs to class fcls0 + (j >> 1)
    auto tap_of = [&](int j) -> int {
        if constexpr (FOLD) {
            const int fcls = fcls0 + (j >> 1);           // (px, py, pz) = bits 2, 1, 0
            const int tb = (wave >> 1) + 4 * (j & 1);    // (sx, sy, sz) = bits 2, 1, 0
            return ((((fcls >> 2) & 1) + ((tb >> 2) & 1)) * 3 + ((fcls >> 1) & 1) + ((tb >> 1) & 1)) * 3 + (fcls & 1) + (tb & 1);
        } else {
            return (COT == 2) ? (wave >> 1) + 4 * j : wave + 8 * j;
        }
    };
    const int ch0 = slice * 32;
    const bool first = ch0 < p.C0;
    const float* src = first ? p.in0 : p.in1;
    const int Cs = first ? p.C0 : p.C1;
    const int chs = first ? ch0 : ch0 - p.C0;
    const bool up = first && p.up0;
    const int X2 = p.X >> 1, Y2 = p.Y >> 1, Z2 = p.Z >> 1;
    constexpr int A_IT = (W_HROWS * 4 + W_THREADS - 1) / W_THREADS;  // 5 items of 8 channels (32 B fp32)
    constexpr int B_IT = 256 * 16 / W_THREADS;                       // 8 items of 4 couts (16 B fp32)

    // lane roles for the transposing reads
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, h = g >> 1, cg = g & 1;
    const int cot = (COT == 2) ? (wave & 1) : 0;
    f32x16 acc[NU];
    int offA_hi[NU], offA_lo[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        int tap = tap_of(j);
        if (tap > 26) tap = 26;
        const int dz = tap % 3;
        const int sw = ((q + dz) >> 1) & 1;  // plane swizzle of the rows this lane addresses (hz = 4r + q + dz)
        const int tapoff = ((tap / 9) * (W_HY * W_HZ) + ((tap / 3) % 3) * W_HZ + dz) * 128;
        offA_hi[j] = tapoff + ((0 ^ sw) << 6);
        offA_lo[j] = tapoff + ((1 ^ sw) << 6);
    }
    int laneA[2], laneBh[2], laneBl[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        laneA[r] = (h * W_HZ + 4 * r + q) * 128 + (cg * 16 + 4 * pp) * 2;
        const int rowb = (8 * h + 4 * r + q) * 256 + (cg * 16 + 4 * pp) * 2;
        laneBh[r] = rowb + (((0 * 2 + cot) ^ q) << 6);
        laneBl[r] = rowb + (((1 * 2 + cot) ^ q) << 6);
    }

    auto tile_origin = [&](int tile, int& b, int& x0, int& y0, int& z0) {
        int t = tile;
        const int tzi = t % p.ntz; t /= p.ntz;
        const int tyi = t % p.nty; t /= p.nty;
        const int txi = t % p.ntx;
        b = t / p.ntx;
        x0 = txi * W_TX; y0 = tyi * W_TY; z0 = tzi * W_TZ;
    };
    // prefetched operands stay native 128-bit vectors from the load to the hi/lo split: assembled element-wise (float4 is a
    // struct) the allocator gave the components of the loop-carried values different registers at the loop header and in
    // the loop body, and the v_mov copies it put behind the loads to reconcile them waited for every load BEFORE the
    // k-loop (s_waitcnt vmcnt(7) ... vmcnt(0)): the prefetch hid nothing
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    struct XItem { u32x4_t a, b; };
    // Branch-free prefetch: every lane loads from a clamped (always in-bounds) address and the value is zeroed with a
    // bit mask afterwards.  With `if (in bounds) load` hipcc branches around every load and waits for it before the
    // next one: the 13 loads of a tile became 13 dependent round trips, 16 k of the 45 k cycles a tile took
    // (tools/wgrad_stamps.py), all of them with the matrix cores idle.
    auto mask4 = [](u32x4_t v, unsigned m) -> u32x4_t { return v & m; };
    unsigned okbits = 0;   // bit it: item `it` of the prefetched X tile is in bounds; bit 16 + it: the same for dZ
    // halo coordinates of this thread's X items, packed once: hx | hy << 4 | hz << 8 | valid << 12 (the row -> (hx, hy, hz)
    // decomposition needs two integer divisions per item; everything derived from it is recomputed per tile, see opaque())
    int xpk[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int i0 = tid + it * W_THREADS;
        const int row = (i0 < W_HROWS * 4 ? i0 : W_HROWS * 4 - 1) >> 2;
        xpk[it] = (row / (W_HY * W_HZ)) | (((row / W_HZ) % W_HY) << 4) | ((row % W_HZ) << 8) | ((i0 < W_HROWS * 4 ? 1 : 0) << 12);
    }
    auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
    // BUF: voxel offset of each X item relative to the tile origin, once per thread (v_mul_lo_u32 runs at a quarter of the
    // VALU rate and the per-item index chain had six of them: 3.5 k cycles per tile went into issuing 13 loads)
    int xdelta[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int hx = xpk[it] & 15, hy = (xpk[it] >> 4) & 15, hz = (xpk[it] >> 8) & 15;
        xdelta[it] = up ? (((hx - 1) >> 1) * Y2 + ((hy - 1) >> 1)) * Z2 + ((hz - 1) >> 1) : ((hx - 1) * p.Y + (hy - 1)) * p.Z + (hz - 1);
    }
    // BUF: operands smaller than 3.75 GB are read through buffer descriptors: an out-of-volume halo item gets the
    // offset 0xF0000000, beyond num_records, and the hardware returns zeros -- no clamping, no masks, 32-bit offsets.
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(src), 0, BUF ? (int)p.bytes_src[first ? 0 : 1] : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dz), 0, BUF ? (int)p.bytes_dz : 0, 0x00020000);
    auto load_x = [&](int tv, int b, int x0, int y0, int z0, int it) -> XItem {
        const int pk = opaque(xpk[it]);
        const int c = tv & 3;
        const int hx = pk & 15, hy = (pk >> 4) & 15, hz = (pk >> 8) & 15;
        const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
        const bool ok = ((pk >> 12) & 1) && gx >= 0 && gx < p.X && gy >= 0 && gy < p.Y && gz >= 0 && gz < p.Z;
        XItem val;
        if constexpr (BUF) {
            // tile origin (wave-uniform, scalar unit) + this item's precomputed delta; x0, y0, z0 are even
            const int org = up ? ((b * X2 + (x0 >> 1)) * Y2 + (y0 >> 1)) * Z2 + (z0 >> 1) : ((b * p.X + x0) * p.Y + y0) * p.Z + z0;
            const unsigned vox = (unsigned)(org + opaque(xdelta[it]));
            const unsigned off = ok ? (vox * (unsigned)Cs + (unsigned)(chs + c * 8)) * 4u : 0xF0000000u;
            val.a = __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0);
            val.b = __builtin_amdgcn_raw_buffer_load_b128(rsX, off + 16u, 0, 0);
            okbits |= 1u << it;   // zeros already delivered by the range check
        } else {
            const int cx = min(max(gx, 0), p.X - 1), cy = min(max(gy, 0), p.Y - 1), cz = min(max(gz, 0), p.Z - 1);
            size_t vox;
            if (up) vox = (((size_t)b * X2 + (cx >> 1)) * Y2 + (cy >> 1)) * Z2 + (cz >> 1);
            else vox = (((size_t)b * p.X + cx) * p.Y + cy) * p.Z + cz;
            const float* qsrc = src + vox * Cs + chs + c * 8;
            okbits = ok ? (okbits | (1u << it)) : (okbits & ~(1u << it));
            // raw: the mask is applied in store_x, AFTER the k-loop -- a use here would make the wave wait for the load
            val.a = *reinterpret_cast<const u32x4_t*>(qsrc);
            val.b = *reinterpret_cast<const u32x4_t*>(qsrc + 4);
        }
        return val;
    };
    // fp32 -> (hi, lo) split in registers (convert_*) and the LDS writes (store_*), both after the top-of-tile barrier.
    // Converting BEFORE the barrier (inside the ~7 k cycles the older half of the workgroup waits there for the younger one)
    // was measured: the converted values stay live across the barrier, the kernel spills again (52 B) and every spill
    // reload between the prefetch loads is a vmcnt(0): 2.58 -> 2.71 ms.
    auto convert_x = [&](int it, const XItem& vraw) -> XItem {   // a = 8 hi halves, b = 8 lo halves (packed pairs)
        XItem v = vraw;
        if constexpr (!BUF) {
            const unsigned m = ((okbits >> it) & 1u) ? 0xffffffffu : 0u;
            v.a = mask4(vraw.a, m);
            v.b = mask4(vraw.b, m);
        }
        const float f[8] = {__uint_as_float(v.a.x), __uint_as_float(v.a.y), __uint_as_float(v.a.z), __uint_as_float(v.a.w),
                            __uint_as_float(v.b.x), __uint_as_float(v.b.y), __uint_as_float(v.b.z), __uint_as_float(v.b.w)};
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            unsigned short h0, l0, h1, l1;
            split_bf16(f[2 * e2], h0, l0);
            split_bf16(f[2 * e2 + 1], h1, l1);
            hi[e2] = (unsigned)h0 | ((unsigned)h1 << 16);
            lo[e2] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
        XItem o;
        o.a = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
        o.b = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
        return o;
    };
    auto store_x = [&](int tv, int it, const XItem& v) {          // v = convert_x(...)
        const int i = tv + it * W_THREADS;
        if (i < W_HROWS * 4) {
            const int row = i >> 2, c = i & 3;
            const int hz = (opaque(xpk[it]) >> 8) & 15;
            const int sw = (hz >> 1) & 1;
            *reinterpret_cast<u32x4_t*>(sX + row * 128 + ((0 ^ sw) << 6) + c * 16) = v.a;
            if constexpr (LO) *reinterpret_cast<u32x4_t*>(sX + row * 128 + ((1 ^ sw) << 6) + c * 16) = v.b;
        }
    };
    auto load_z = [&](int tv, int b, int x0, int y0, int z0, int it, int fcls = 0) -> u32x4_t {
        const int i = tv + it * W_THREADS;
        const int v = i >> 4, c4 = i & 15;
        const int gx = x0 + (v >> 6), gy = y0 + ((v >> 3) & 7), gz = z0 + (v & 7);
        const int co = cob * 64 + c4 * 4;
        if constexpr (FOLD) {
            // low-resolution voxel (gx, gy, gz) of the tile <-> voxel 2 g + p of the full-resolution dz [B, 2X, 2Y, 2Z, Cout]
            const bool ok = gx < p.X && gy < p.Y && gz < p.Z;
            const unsigned vox = (unsigned)(((b * 2 * p.X + 2 * gx + ((fcls >> 2) & 1)) * 2 * p.Y + 2 * gy + ((fcls >> 1) & 1)) * 2 * p.Z +
                                            2 * gz + (fcls & 1));
            okbits |= 0x10000u << it;
            return __builtin_amdgcn_raw_buffer_load_b128(rsZ, ok ? (vox * (unsigned)p.Cout + (unsigned)co) * 4u : 0xF0000000u, 0, 0);
        } else if constexpr (FULLCO && BUF) {
            // item `it` = voxel (vx = it >> 1, vy = vy0 + 4 (it & 1), vz): one per-thread base offset, the item part is
            // wave-uniform and goes into the instruction's scalar offset
            const int v0 = tv >> 4, vy0 = (v0 >> 3) & 3, vz0 = v0 & 7;
            const bool ok = (x0 + (it >> 1)) < p.X && (y0 + vy0 + 4 * (it & 1)) < p.Y && (z0 + vz0) < p.Z;
            const int org = ((b * p.X + x0) * p.Y + y0) * p.Z + z0;
            const unsigned off0 = ((unsigned)(org + vy0 * p.Z + vz0) * (unsigned)p.Cout + (unsigned)co) * 4u;
            const int soff = (((it >> 1) * p.Y + 4 * (it & 1)) * p.Z) * p.Cout * 4;
            okbits |= 0x10000u << it;
            return __builtin_amdgcn_raw_buffer_load_b128(rsZ, ok ? off0 : 0xF0000000u, soff, 0);
        } else if constexpr (FULLCO) {   // branch-free (see load_x): clamped address, bit mask applied at store time
            const bool ok = gx < p.X && gy < p.Y && gz < p.Z;
            const int cx = min(gx, p.X - 1), cy = min(gy, p.Y - 1), cz = min(gz, p.Z - 1);
            const float* qq = p.dz + ((((size_t)b * p.X + cx) * p.Y + cy) * p.Z + cz) * p.Cout;
            okbits = ok ? (okbits | (0x10000u << it)) : (okbits & ~(0x10000u << it));
            return *reinterpret_cast<const u32x4_t*>(qq + co);   // raw; masked in store_z
        } else {
        okbits |= 0x10000u << it;
        u32x4_t val = {0u, 0u, 0u, 0u};
        if (gx < p.X && gy < p.Y && gz < p.Z) {
            const float* qq = p.dz + ((((size_t)b * p.X + gx) * p.Y + gy) * p.Z + gz) * p.Cout;
            if (co + 3 < p.Cout) val = *reinterpret_cast<const u32x4_t*>(qq + co);
            else {
                if (co < p.Cout) val.x = __float_as_uint(qq[co]);
                if (co + 1 < p.Cout) val.y = __float_as_uint(qq[co + 1]);
                if (co + 2 < p.Cout) val.z = __float_as_uint(qq[co + 2]);
            }
        }
        return val;
        }
    };
    auto convert_z = [&](int it, u32x4_t vraw) -> u32x4_t {       // (x, y) = 4 hi halves, (z, w) = 4 lo halves
        const u32x4_t vm = (FULLCO && BUF) ? vraw : mask4(vraw, ((okbits >> (16 + it)) & 1u) ? 0xffffffffu : 0u);
        unsigned short h0, l0, h1, l1, h2, l2, h3, l3;
        split_bf16(__uint_as_float(vm.x), h0, l0); split_bf16(__uint_as_float(vm.y), h1, l1);
        split_bf16(__uint_as_float(vm.z), h2, l2); split_bf16(__uint_as_float(vm.w), h3, l3);
        return u32x4_t{(unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16),
                       (unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16)};
    };
    auto store_z = [&](int tv, int it, u32x4_t v) {               // v = convert_z(...)
        const int i = tv + it * W_THREADS;
        const int vv = i >> 4, c4 = i & 15;
        const int vz = vv & 7;
        const int seg_hi = (0 * 2 + (c4 >> 3)) ^ (vz & 3), seg_lo = (1 * 2 + (c4 >> 3)) ^ (vz & 3);
        *reinterpret_cast<uint2*>(sZ + vv * 256 + (seg_hi << 6) + (c4 & 7) * 8) = make_uint2(v.x, v.y);
        if constexpr (LO) *reinterpret_cast<uint2*>(sZ + vv * 256 + (seg_lo << 6) + (c4 & 7) * 8) = make_uint2(v.z, v.w);
    };

    XItem px[A_IT];
    u32x4_t pz[B_IT];
    // `opaque_tid()`: the staging index arithmetic (row -> halo coordinates -> clamped address, 13 items per thread) is
    // invariant across voxel tiles, so hipcc hoists all of it out of the tile loop and keeps it live across the k-loop,
    // next to 112 accumulator registers: it is what spilled.  An empty asm makes the thread index opaque per use, so
    // the arithmetic (a few hundred VALU per 45 k-cycle tile) is redone where it is needed and nothing stays live.
    auto opaque_tid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    int pass = 0, nxt = -1, tile = xcd_tile((int)blockIdx.x, (int)gridDim.x, 0, p.ntiles);      // (see wgrad_kernel)
    if constexpr (FOLD) {
        // Four classes per workgroup and voxel tile: the x_low tile is staged once, then per class its dz sub-lattice tile and a
        // k-loop over the class's two accumulator tiles.  The next operand tile (dz of the next class, or x_low + dz of the next
        // voxel tile) is prefetched into registers under each k-loop.
        if (tile >= 0) {
            const int tv = opaque_tid();
            int b, x0, y0, z0;
            tile_origin(tile, b, x0, y0, z0);
#pragma unroll
            for (int it = 0; it < A_IT; ++it) px[it] = load_x(tv, b, x0, y0, z0, it);
#pragma unroll
            for (int it = 0; it < B_IT; ++it) pz[it] = load_z(tv, b, x0, y0, z0, it, fcls0);
        }
        for (; tile >= 0; tile = nxt, ++pass) {
            int b, x0, y0, z0;
            tile_origin(tile, b, x0, y0, z0);
            nxt = xcd_tile((int)blockIdx.x, (int)gridDim.x, pass + 1, p.ntiles);
            int nb = 0, nx0 = 0, ny0 = 0, nz0 = 0;
            if (nxt >= 0) tile_origin(nxt, nb, nx0, ny0, nz0);
            __syncthreads();          // the previous tile's last k-loop is done with sX and sZ
            {
                const int tv = opaque_tid();
#pragma unroll
                for (int it = 0; it < A_IT; ++it) store_x(tv, it, convert_x(it, px[it]));
            }
            auto one_class = [&](auto cc) {
                constexpr int C = decltype(cc)::value;
                if constexpr (C > 0) __syncthreads();     // class C - 1's k-loop is done with sZ
                {
                    const int tv = opaque_tid();
#pragma unroll
                    for (int it = 0; it < B_IT; ++it) store_z(tv, it, convert_z(it, pz[it]));
                }
                __syncthreads();
                {
                    const int tv2 = opaque_tid();
                    if constexpr (C + 1 < FNC) {
#pragma unroll
                        for (int it = 0; it < B_IT; ++it) pz[it] = load_z(tv2, b, x0, y0, z0, it, fcls0 + C + 1);
                    } else if (nxt >= 0) {
#pragma unroll
                        for (int it = 0; it < A_IT; ++it) px[it] = load_x(tv2, nb, nx0, ny0, nz0, it);
#pragma unroll
                        for (int it = 0; it < B_IT; ++it) pz[it] = load_z(tv2, nb, nx0, ny0, nz0, it, fcls0);
                    }
                }
#pragma unroll 1
                for (int kb = 0; kb < 16; ++kb) {
                    const char* xa = sX + ((kb >> 2) * (W_HY * W_HZ) + (kb & 3) * 2 * W_HZ) * 128;
                    const char* zb = sZ + kb * 16 * 256;
                    const bf16x8_t b_hi = frag8(tr_read(zb + laneBh[0]), tr_read(zb + laneBh[1]));
                    bf16x8_t b_lo = b_hi;
                    if constexpr (LO) b_lo = frag8(tr_read(zb + laneBl[0]), tr_read(zb + laneBl[1]));
#pragma unroll
                    for (int j = 2 * C; j < 2 * C + 2; ++j) {
                        const bf16x8_t a_hi = frag8(tr_read(xa + laneA[0] + offA_hi[j]), tr_read(xa + laneA[1] + offA_hi[j]));
                        if constexpr (LO) {
                            const bf16x8_t a_lo = frag8(tr_read(xa + laneA[0] + offA_lo[j]), tr_read(xa + laneA[1] + offA_lo[j]));
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[j], 0, 0, 0);
                        }
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[j], 0, 0, 0);
                    }
                }
            };
            one_class(IC<0>{}); one_class(IC<1>{}); one_class(IC<2>{}); one_class(IC<3>{});
        }
    } else {
    if (tile >= 0) {
        const int tv = opaque_tid();
        int b, x0, y0, z0;
        tile_origin(tile, b, x0, y0, z0);
        if constexpr (PFX) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) px[it] = load_x(tv, b, x0, y0, z0, it);
        }
        if constexpr (PFZ) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) pz[it] = load_z(tv, b, x0, y0, z0, it);
        }
    }
    for (; tile >= 0; tile = nxt, ++pass) {
        if constexpr (STAMP) st_t = wstamp_now();
        __syncthreads();
        if constexpr (STAMP) { const unsigned long long t = wstamp_now(); st_acc[0] += t - st_t; st_t = t; }
        const int tv = opaque_tid();
        if constexpr (!PFX || !PFZ) {   // the operand tile that is not prefetched: all its loads in one batch, then the stores
            int b, x0, y0, z0;
            tile_origin(tile, b, x0, y0, z0);
            if constexpr (!PFX) {
#pragma unroll
                for (int it = 0; it < A_IT; ++it) px[it] = load_x(tv, b, x0, y0, z0, it);
            }
            if constexpr (!PFZ) {
#pragma unroll
                for (int it = 0; it < B_IT; ++it) pz[it] = load_z(tv, b, x0, y0, z0, it);
            }
        }
#pragma unroll
        for (int it = 0; it < A_IT; ++it) store_x(tv, it, convert_x(it, px[it]));
#pragma unroll
        for (int it = 0; it < B_IT; ++it) store_z(tv, it, convert_z(it, pz[it]));
        __syncthreads();
        if constexpr (STAMP) { const unsigned long long t = wstamp_now(); st_acc[1] += t - st_t; st_t = t; }
        nxt = xcd_tile((int)blockIdx.x, (int)gridDim.x, pass + 1, p.ntiles);
        // the next tile's loads go out in one burst in front of the k-loop (spread over its quarters they compete with the
        // transposing LDS reads for the VGPR write path: same 2.6 ms per launch)
        int nb = 0, nx0 = 0, ny0 = 0, nz0 = 0;
        if (nxt >= 0) tile_origin(nxt, nb, nx0, ny0, nz0);
        auto issue_group = [&](auto gc) {
            constexpr int GQ = decltype(gc)::value;          // 0, 1: X items; 2, 3: dZ items
            if (nxt >= 0) {
                const int tv2 = opaque_tid();
                if constexpr (GQ < 2 && PFX) {
                    constexpr int A0 = GQ == 0 ? 0 : (A_IT + 1) / 2, A1 = GQ == 0 ? (A_IT + 1) / 2 : A_IT;
#pragma unroll
                    for (int it = A0; it < A1; ++it) px[it] = load_x(tv2, nb, nx0, ny0, nz0, it);
                }
                if constexpr (GQ >= 2 && PFZ) {
                    constexpr int B0 = GQ == 2 ? 0 : (B_IT + 1) / 2, B1 = GQ == 2 ? (B_IT + 1) / 2 : B_IT;
#pragma unroll
                    for (int it = B0; it < B1; ++it) pz[it] = load_z(tv2, nb, nx0, ny0, nz0, it);
                }
            }
        };
        issue_group(IC<0>{}); issue_group(IC<1>{}); issue_group(IC<2>{}); issue_group(IC<3>{});
        if constexpr (STAMP) { const unsigned long long t = wstamp_now(); st_acc[2] += t - st_t; st_t = t; }
        auto k_range = [&](int k0, int k1) {
#pragma unroll 1
        for (int kb = k0; kb < k1; ++kb) {  // 16 voxels per k-block: (vx = kb>>2, vy = (kb&3)*2 + {0,1}, vz = 0..7)
            const char* xa = sX + ((kb >> 2) * (W_HY * W_HZ) + (kb & 3) * 2 * W_HZ) * 128;
            const char* zb = sZ + kb * 16 * 256;
            const bf16x8_t b_hi = frag8(tr_read(zb + laneBh[0]), tr_read(zb + laneBh[1]));
            bf16x8_t b_lo = b_hi;
            if constexpr (LO) b_lo = frag8(tr_read(zb + laneBl[0]), tr_read(zb + laneBl[1]));
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const bf16x8_t a_hi = frag8(tr_read(xa + laneA[0] + offA_hi[j]), tr_read(xa + laneA[1] + offA_hi[j]));
                if constexpr (LO) {
                    const bf16x8_t a_lo = frag8(tr_read(xa + laneA[0] + offA_lo[j]), tr_read(xa + laneA[1] + offA_lo[j]));
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[j], 0, 0, 0);
                }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[j], 0, 0, 0);
            }
            if constexpr (LO && COT == 2) {
                // fragment pipeline: dZ + the first two X units' reads up front, then each unit's three MFMAs with the
                // reads of the unit two ahead behind them (hipcc alone waits lgkmcnt(0) in front of every unit)
                __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#define MMR_WG(rd) __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); if (rd) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0)
                MMR_WG(1); MMR_WG(1); MMR_WG(1); MMR_WG(1); MMR_WG(1); MMR_WG(0); MMR_WG(0);
#undef MMR_WG
            }
        }
        };
        k_range(0, 16);
        if constexpr (STAMP) { const unsigned long long t = wstamp_now(); st_acc[3] += t - st_t; st_acc[4] += 1; }
    }
    }   // !FOLD
    if constexpr (STAMP) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) atomicAdd(&g_wgrad_stamp[wave][k], st_acc[k]);
        }
    }
    // slab[blk][slice][cob (FOLD: cob * 8 + class)][27 taps][32 ci][64 co]
    const int zcount = FOLD ? (int)gridDim.z * FNC : (int)gridDim.z;     // FOLD: gridDim.z = 2 cob blocks of 4 classes each
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const int zi = FOLD ? cob * 8 + fcls0 + (j >> 1) : (int)blockIdx.z;
        float* out = p.slab + (((size_t)blockIdx.x * gridDim.y + slice) * zcount + zi) * (27 * 32 * 64);
        const int tap = tap_of(j);
        if (tap < 27) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(tap * 32 + ci) * 64 + cot * 32 + (lane & 31)] = acc[j][r];
            }
        }
    }
}

// dW[tap][ci_off + ci][co] (+)= sum over slabs (fixed order); dW has cin_total input-channel rows per tap (a layer's whole
// Keras kernel), the slabs cover the Cin channels starting at ci_off
__global__ void __launch_bounds__(TB)
wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nblk, int nslices, int ncob, int Cin,
                    int Cout, int accumulate, int cin_total, int ci_off)
{
    const int64_t total = (int64_t)27 * Cin * Cout;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TB) {
        const int co = (int)(i % Cout);
        const int ci = (int)((i / Cout) % Cin);
        const int tap = (int)(i / ((int64_t)Cout * Cin));
        const int slice = ci >> 5, cob = co >> 6;
        // eight slabs in flight per thread (the one-accumulator form ran as nblk dependent round trips: 23 us per layer at
        // 160^3); fixed association -> still bitwise reproducible
        const float* base = slab + (((size_t)slice * ncob + cob) * 27 + tap) * 2048 + (ci & 31) * 64 + (co & 63);
        const size_t kstride = (size_t)nslices * ncob * 27 * 2048;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 7 < nblk; k += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += base[(size_t)(k + u) * kstride];
        }
        for (; k < nblk; ++k) a[0] += base[(size_t)k * kstride];
        const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        float* o = dw + ((int64_t)tap * cin_total + ci_off + ci) * Cout + co;
        if (accumulate) *o += s; else *o = s;
    }
}

// Folded wgrad: dW[t][ci][co] (+)= sum over slabs and over the 8 (class, tap bit) pairs whose fold covers original tap t --
// per axis t = 0: (p, s) = (0, 0), (1, 0); t = 1: (0, 1), (1, 0); t = 2: (0, 1), (1, 1) -- read from slab entry (class p,
// 27-index p + s).  Fixed summation order; the slabs cover the first C0 input channels of a cin_total-row kernel.
__global__ void __launch_bounds__(TB)
wgrad_fold_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nblk, int nslices, int ncob, int C0,
                         int Cout, int accumulate, int cin_total)
{
    const int64_t total = (int64_t)27 * C0 * Cout;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TB) {
        const int co = (int)(i % Cout);
        const int ci = (int)((i / Cout) % C0);
        const int tap = (int)(i / ((int64_t)Cout * C0));
        const int slice = ci >> 5, cob = co >> 6;
        const int t3[3] = {tap / 9, (tap / 3) % 3, tap % 3};
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // one accumulator per (class, tap bit) pair: 8 loads in flight
        for (int k = 0; k < nblk; ++k) {
            const float* base = slab + ((((size_t)k * nslices + slice) * ncob + cob) * 8) * (27 * 2048) + (ci & 31) * 64 + (co & 63);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                int cls = 0, tau = 0;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int pick = (m >> (2 - a)) & 1, t = t3[a];
                    // the two (p, s) pairs covering t on this axis
                    const int pp = t == 0 ? pick : (t == 1 ? pick : pick);            // p = 0 or 1
                    const int ss = t == 0 ? 0 : (t == 1 ? 1 - pick : 1);              // t=0: s=0; t=1: (0,1),(1,0); t=2: s=1
                    cls = (cls << 1) | pp;
                    tau = tau * 3 + pp + ss;
                }
                a8[m] += base[((size_t)cls * 27 + tau) * 2048];
            }
        }
        const float s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        float* o = dw + ((int64_t)tap * cin_total + ci) * Cout + co;
        if (accumulate) *o += s; else *o = s;
    }
}

// first layer wgrad (Cin = 2): dW[27][2][Cout]; thread = (co, voxel group); X patch in LDS
constexpr int G_TX = 4, G_TY = 4, G_TZ = 16;
__global__ void __launch_bounds__(TB)
wgrad_cin2_kernel(const float* __restrict__ src, const float* __restrict__ trg, const float* __restrict__ dz,
                  float* __restrict__ part, int B, int X, int Y, int Z, int Cout, int ntx, int nty, int ntz, int ntiles)
{
    constexpr int PX = G_TX + 2, PY = G_TY + 2, PZ = G_TZ + 2;
    __shared__ float patch[PX][PY][PZ][2];
    __shared__ float red[TB];
    const int ngrp = TB / Cout;  // Cout in {32,64,128,256}
    const int co = threadIdx.x % Cout, grp = threadIdx.x / Cout;
    float acc[54];
#pragma unroll
    for (int k = 0; k < 54; ++k) acc[k] = 0.f;
    const size_t nvox = (size_t)X * Y * Z;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tzi = t % ntz; t /= ntz;
        const int tyi = t % nty; t /= nty;
        const int txi = t % ntx;
        const int b = t / ntx;
        const int x0 = txi * G_TX, y0 = tyi * G_TY, z0 = tzi * G_TZ;
        __syncthreads();
        for (int i = threadIdx.x; i < PX * PY * PZ; i += TB) {
            const int hz = i % PZ, hy = (i / PZ) % PY, hx = i / (PZ * PY);
            const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
            float a = 0.f, c = 0.f;
            if (gx >= 0 && gx < X && gy >= 0 && gy < Y && gz >= 0 && gz < Z) {
                const size_t o = (size_t)b * nvox + ((size_t)gx * Y + gy) * Z + gz;
                a = src[o];
                c = trg[o];
            }
            patch[hx][hy][hz][0] = a;
            patch[hx][hy][hz][1] = c;
        }
        __syncthreads();
        for (int v = grp; v < G_TX * G_TY * G_TZ; v += ngrp) {
            const int vz = v % G_TZ, vy = (v / G_TZ) % G_TY, vx = v / (G_TZ * G_TY);
            const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
            if (gx < X && gy < Y && gz < Z) {
                const float g = dz[((size_t)b * nvox + ((size_t)gx * Y + gy) * Z + gz) * Cout + co];
#pragma unroll
                for (int k = 0; k < 27; ++k) {
                    const float2 pv = *reinterpret_cast<const float2*>(&patch[vx + k / 9][vy + (k / 3) % 3][vz + k % 3][0]);
                    acc[2 * k] = fmaf(pv.x, g, acc[2 * k]);
                    acc[2 * k + 1] = fmaf(pv.y, g, acc[2 * k + 1]);
                }
            }
        }
    }
    // reduce the voxel groups, write the block partial [54][Cout]
    for (int k = 0; k < 54; ++k) {
        __syncthreads();
        red[threadIdx.x] = acc[k];
        __syncthreads();
        if (grp == 0) {
            float s = 0.f;
            for (int g = 0; g < ngrp; ++g) s += red[g * Cout + co];
            part[((size_t)blockIdx.x * 54 + k) * Cout + co] = s;
        }
    }
}

__global__ void __launch_bounds__(TB)
sum_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n, int nblk, int accumulate)
{
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        float s = 0.f;
        for (int k = 0; k < nblk; ++k) s += part[(int64_t)k * n + i];
        if (accumulate) out[i] += s; else out[i] = s;
    }
}

// flow head dgrad: dX[v][ci] = sum_tap sum_co dY[v - off(tap)][co] * W[tap][ci][co]   (Cout = 3)
// Lanes = voxels: each lane gathers its 27 x 3 dY neighbourhood into registers once, then the wave loops
// over the input channels with the 81 weights of that channel wave-uniform (scalar loads, K$-resident:
// 27*Cin*3 floats), 81 v_fma per channel, 16-B stores of 4 channels per lane (a lane writes its whole
// Cin row over the loop, so L2 merges the pieces into full lines).
__global__ void __launch_bounds__(TB)
dgrad_cout3_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int B, int X,
                   int Y, int Z, int Cin)
{
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t total = (int64_t)B * nvox;
    const int64_t v = (int64_t)blockIdx.x * TB + threadIdx.x;
    const bool live = v < total;
    const int64_t vv = live ? v : 0;
    const int64_t b = vv / nvox;
    const int64_t r = vv - b * nvox;
    const int z = (int)(r % Z), y = (int)((r / Z) % Y), x = (int)(r / ((int64_t)Z * Y));
    float g[81];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        const int sx = x - (t / 9 - 1), sy = y - ((t / 3) % 3 - 1), sz = z - (t % 3 - 1);
        const bool ok = sx >= 0 && sx < X && sy >= 0 && sy < Y && sz >= 0 && sz < Z;
        const float* q = dy + (b * nvox + ((int64_t)(ok ? sx : x) * Y + (ok ? sy : y)) * Z + (ok ? sz : z)) * 3;
        g[t * 3] = ok ? q[0] : 0.f;
        g[t * 3 + 1] = ok ? q[1] : 0.f;
        g[t * 3 + 2] = ok ? q[2] : 0.f;
    }
    float* o = dx + vv * Cin;
    for (int c4 = 0; c4 < Cin; c4 += 4) {
        float a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* ww = w + (size_t)(c4 + k) * 3;  // w[(t*Cin + ci)*3 + co], wave-uniform
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const float* wt = ww + (size_t)t * Cin * 3;
                acc = fmaf(g[t * 3], wt[0], acc);
                acc = fmaf(g[t * 3 + 1], wt[1], acc);
                acc = fmaf(g[t * 3 + 2], wt[2], acc);
            }
            a[k] = acc;
        }
        if (live) *reinterpret_cast<float4*>(o + c4) = make_float4(a[0], a[1], a[2], a[3]);
    }
}

// flow head wgrad: dW[t][ci][co] = sum_v X[v+off(t)][ci] * dY[v][co]  (Cout = 3).  Lanes = input channels;
// a wave walks z-runs of voxels: X rows are coalesced 256-B loads (27x reuse through L1), dY[v][0..2] is
// wave-uniform (scalar loads), 81 accumulators per lane; per-block partials are reduced in fixed order.
constexpr int WG3_RUN = 32;
__global__ void __launch_bounds__(TB)
wgrad_cout3_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int B, int X,
                   int Y, int Z, int Cin, int nzr, int64_t njobs)
{
    __shared__ float red[TB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ncb = Cin / 64;
    const int cb = blockIdx.y;
    const int ci = cb * 64 + lane;
    float acc[81];
#pragma unroll
    for (int k = 0; k < 81; ++k) acc[k] = 0.f;
    const int64_t nvox = (int64_t)X * Y * Z;
    for (int64_t job = (int64_t)blockIdx.x * (TB / 64) + wave; job < njobs; job += (int64_t)gridDim.x * (TB / 64)) {
        int64_t j = job;
        const int zr = (int)(j % nzr); j /= nzr;
        const int y = (int)(j % Y); j /= Y;
        const int xx = (int)(j % X);
        const int b = (int)(j / X);
        const int z0 = zr * WG3_RUN;
        const int z1 = (z0 + WG3_RUN < Z) ? z0 + WG3_RUN : Z;
        const float* xb = x + (size_t)b * nvox * Cin + ci;
        const float* dyb = dy + (size_t)b * nvox * 3;
        for (int z = z0; z < z1; ++z) {
            const float* gq = dyb + (((size_t)xx * Y + y) * Z + z) * 3;
            const float g0 = gq[0], g1 = gq[1], g2 = gq[2];
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int sx = xx + (t / 9 - 1), sy = y + ((t / 3) % 3 - 1), sz = z + (t % 3 - 1);
                const bool ok = sx >= 0 && sx < X && sy >= 0 && sy < Y && sz >= 0 && sz < Z;  // wave-uniform
                const float xv = ok ? xb[(((size_t)sx * Y + sy) * Z + sz) * Cin] : 0.f;
                acc[t * 3] = fmaf(xv, g0, acc[t * 3]);
                acc[t * 3 + 1] = fmaf(xv, g1, acc[t * 3 + 1]);
                acc[t * 3 + 2] = fmaf(xv, g2, acc[t * 3 + 2]);
            }
        }
    }
    // combine the block's 4 waves, write partial [blk][27][Cin][3]
    for (int k = 0; k < 81; ++k) {
        __syncthreads();
        red[threadIdx.x] = acc[k];
        __syncthreads();
        if (wave == 0) {
            const float sum = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
            part[((size_t)blockIdx.x * 27 + k / 3) * Cin * 3 + (size_t)ci * 3 + (k % 3)] = sum;
        }
    }
    (void)ncb;
}

// generic fallback (any Cin): one thread per (voxel, ci), weights in LDS
__global__ void __launch_bounds__(TB)
dgrad_cout3_generic_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int B,
                           int X, int Y, int Z, int Cin)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sw = reinterpret_cast<float*>(smem);  // [27][Cin][3]
    for (int i = threadIdx.x; i < 27 * Cin * 3; i += TB) sw[i] = w[i];
    __syncthreads();
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t total = (int64_t)B * nvox * Cin;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TB) {
        const int ci = (int)(i % Cin);
        const int64_t v = i / Cin;
        const int64_t b = v / nvox;
        const int64_t r = v - b * nvox;
        const int z = (int)(r % Z), y = (int)((r / Z) % Y), x = (int)(r / ((int64_t)Z * Y));
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int sxp = x - (tap / 9 - 1), syp = y - ((tap / 3) % 3 - 1), szp = z - (tap % 3 - 1);
            if (sxp >= 0 && sxp < X && syp >= 0 && syp < Y && szp >= 0 && szp < Z) {
                const float* g = dy + (b * nvox + ((int64_t)sxp * Y + syp) * Z + szp) * 3;
                const float* ww = sw + (tap * Cin + ci) * 3;
                acc += g[0] * ww[0] + g[1] * ww[1] + g[2] * ww[2];
            }
        }
        dx[i] = acc;
    }
}

// ------------------------------------------------------------------------- //
// Thin layers on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32)   //
// ------------------------------------------------------------------------- //
// The first layer (Cin = 2) and the flow head (Cout = 3) have one tiny channel axis.  Folding the 27 taps
// into it gives a GEMM dimension of 54 / 81 "virtual channels" G[v][tap*Cs + c] = S[v +- off(tap)][c] that
// is gathered on the fly from a haloed LDS tile of the small tensor S, so no 10x zero padding of the
// channel axis is multiplied.
constexpr int SM_THREADS = 256;

// T[d][g] = sum_v Dn[v][d] * G[v][g];  d < 64 (one 64-channel block of the dense tensor per blockIdx.y),
// g < 27*Cs (<= 96).  sign = +1: G[v][tap,c] = S[v + off(tap)][c] (first-layer wgrad: S = image pair),
// sign = -1: S[v - off(tap)][c] (flow-head wgrad: S = dflow).  part: [gridDim.x*4][gridDim.y][64][96].
__global__ void __launch_bounds__(SM_THREADS, 2)
smallch_wgrad_kernel(const float* __restrict__ dense, int Cd, const float* __restrict__ s0, const float* __restrict__ s1,
                     int Cs, int sign, float* __restrict__ part, int B, int X, int Y, int Z, int ntx, int nty, int ntz,
                     int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sS = reinterpret_cast<float*>(smem);            // [600][Cs]
    float* sD = sS + W_HROWS * 3;                          // [256][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int cb = blockIdx.y;
    const int KG = 27 * Cs;
    int goff[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const int gcol = n * 32 + (lane & 31);
        int o = 0;
        if (gcol < KG) {
            const int tap = gcol / Cs, c = gcol % Cs;
            const int d = ((tap / 9 - 1) * (W_HY * W_HZ) + ((tap / 3) % 3 - 1) * W_HZ + (tap % 3 - 1));
            o = sign * d * Cs + c;
        }
        goff[n] = o;
    }
    f32x16 acc[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const size_t nvox = (size_t)X * Y * Z;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tzi = t % ntz; t /= ntz;
        const int tyi = t % nty; t /= nty;
        const int txi = t % ntx;
        const int b = t / ntx;
        const int x0 = txi * W_TX, y0 = tyi * W_TY, z0 = tzi * W_TZ;
        __syncthreads();
        for (int i = tid; i < W_HROWS; i += SM_THREADS) {
            const int hx = i / (W_HY * W_HZ), hy = (i / W_HZ) % W_HY, hz = i % W_HZ;
            const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
            const bool ok = gx >= 0 && gx < X && gy >= 0 && gy < Y && gz >= 0 && gz < Z;
            const size_t o = (size_t)b * nvox + ((size_t)gx * Y + gy) * Z + gz;
            if (s1) {  // two single-channel volumes (moving, fixed)
                sS[i * Cs] = ok ? s0[o] : 0.f;
                sS[i * Cs + 1] = ok ? s1[o] : 0.f;
            } else {
                for (int c = 0; c < Cs; ++c) sS[i * Cs + c] = ok ? s0[o * Cs + c] : 0.f;
            }
        }
        for (int i = tid; i < 256 * 16; i += SM_THREADS) {
            const int v = i >> 4, c4 = i & 15;
            const int gx = x0 + (v >> 6), gy = y0 + ((v >> 3) & 7), gz = z0 + (v & 7);
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gx < X && gy < Y && gz < Z)
                val = *reinterpret_cast<const float4*>(dense + ((size_t)b * nvox + ((size_t)gx * Y + gy) * Z + gz) * Cd +
                                                       cb * 64 + c4 * 4);
            *reinterpret_cast<float4*>(sD + v * 64 + c4 * 4) = val;
        }
        __syncthreads();
#pragma unroll 4
        for (int st = 0; st < 32; ++st) {
            const int v = wave * 64 + st * 2 + h;
            const int hrow = (((v >> 6) + 1) * W_HY + ((v >> 3) & 7) + 1) * W_HZ + (v & 7) + 1;
            const float a0 = sD[v * 64 + (lane & 31)], a1 = sD[v * 64 + 32 + (lane & 31)];
            float bv[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) bv[n] = sS[hrow * Cs + goff[n]];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[n], acc[0][n], 0, 0, 0);
                acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[n], acc[1][n], 0, 0, 0);
            }
        }
    }
    float* o = part + (((size_t)(blockIdx.x * 4 + wave)) * gridDim.y + cb) * (64 * 96);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                o[d * 96 + n * 32 + (lane & 31)] = acc[m][n][r];
            }
}

// sum the partial T's and scatter into the Keras weight-gradient layout
//   mode 0 (flow head):   dW[tap][d][c]  = T[d][tap*Cs + c]   (d = input channel, Cd of them; Cs = 3)
//   mode 1 (first layer): dW[tap][c][d]  = T[d][tap*Cs + c]   (d = output channel; Cs = 2)
__global__ void __launch_bounds__(64)
smallch_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nparts, int ncb, int Cd, int Cs,
                            int mode, int accumulate)
{
    const int KG = 27 * Cs;
    const int i = blockIdx.x;  // one wave per output element: ordered strided sum + wave reduction
    const int g = i % KG, d = i / KG;
    const int cb = d >> 6, dl = d & 63;
    float s = 0.f;
    for (int k = threadIdx.x; k < nparts; k += 64) s += part[(((size_t)k) * ncb + cb) * (64 * 96) + dl * 96 + g];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        const int tap = g / Cs, c = g % Cs;
        const size_t o = mode == 0 ? ((size_t)tap * Cd + d) * Cs + c : ((size_t)tap * Cs + c) * Cd + d;
        if (accumulate) dw[o] += s; else dw[o] = s;
    }
}

// ------------------------------------------------------------------------- //
// The same thin-layer weight gradient on the bf16 cores (fp32x3)              //
// ------------------------------------------------------------------------- //
// Same GEMM and the same partial layout as smallch_wgrad_kernel (T[d][g] = sum_v Dn[v][d] * G[v][g]), every product as
// three v_mfma_f32_32x32x16_bf16 of bf16 (hi, lo) halves.  The contraction runs over voxels, so both operands need 8
// consecutive voxels per lane: a k-step is two z-rows of 8 voxels and LDS holds, per operand, 16-byte cells of 8
// z-consecutive bf16 -- one ds_read_b128 per fragment, no transposing read:
//   sD [hi|lo][d = 64][z-row = 16] cells, 256 B per d, the z-row XOR-swizzled with tw_dswz(d): conflict-free for the fragment
//      reads (lanes = consecutive d) AND for the staging stores (lanes = every fourth d; with the round-2 layout, +16 B per d and
//      no swizzle, those were 4-way conflicted: 58 % of the kernel's LDS cycles, profiles/r04k_train_pmc_sq.json)
//   sS [hi|lo][c][dz = 3][haloed (x, y) row = 40] cells: the z window of every tap shift is stored as its own
//      aligned cell, so the gather G[v][tap, c] = S[v +- off(tap)][c] is a lane-constant offset + the row.
// Tile = 2 x 8 x 8 voxels (16 z-rows = 8 k-steps, two per wave); the next tile's dense operand (8 x 16 B per thread)
// and S rows are prefetched into registers under the MFMAs; fp32 -> (hi, lo) happens on the way into LDS.  The
// wave partials are summed in a fixed order through LDS: one [64][96] partial per workgroup.
constexpr int TW_TX = 2, TW_TY = 8, TW_TZ = 8;
constexpr int TW_HX = TW_TX + 2, TW_HY = TW_TY + 2;
constexpr int TW_DSTR = 16 * 16;                      // bytes per dense channel
constexpr int TW_DHL = 64 * TW_DSTR;                  // hi -> lo plane
constexpr int TW_D_BYTES = 2 * TW_DHL;                // 32,768
// z-row swizzle of dense channel d.  A ds_read_b128 lane group holds 16 channels with distinct d mod 16 ({0-3, 12-15, 20-27} /
// {4-11, 16-19, 28-31}): pi(d & 15) ^ (bit 4 of d) << 2 with pi = swap of the two 2-bit halves is a bijection onto the 16 slots
// for both groups.  A ds_write_b128 group is 8 lanes with d = 4 k + j (k = 0 .. 7): pi puts k & 3 into the low bits and bit 4 of d
// (= k >> 2) flips bit 2, so the 8 lanes take the 8 slots of the 32 banks.
__host__ __device__ inline int tw_dswz(int d) { return (((d >> 2) & 3) | ((d & 3) << 2)) ^ (((d >> 4) & 1) << 2); }
// Column order of the small operand.  Column position j = n * 32 + li of the three N tiles holds the (tap, c) combination
// g[j] & 127; bit 7 marks a padding column (it re-reads the cell of another lane of its group: a broadcast).  The host deals
// the 27 Cs real columns over the six ds_read_b128 lane groups so that no two lanes of a group fall on the same 4-bank slot
// (in natural order the S fragment reads were 2-way conflicted); the partials are stored back in natural order.
struct TwPerm { unsigned char g[96]; };
constexpr int TW_SPL = TW_HX * TW_HY * 16 + 16;       // bytes per (c, dz) plane of S
constexpr int TW_SHL = 9 * TW_SPL;                    // hi -> lo plane (Cs <= 3)
constexpr int TW_S_BYTES = 2 * TW_SHL;                // 11,808
constexpr int TW_LDS = TW_D_BYTES + TW_S_BYTES;

typedef __attribute__((ext_vector_type(4))) unsigned tw_u32x4;
typedef __attribute__((ext_vector_type(2))) float tw_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 tw_bf16x2;

// (a, b) -> packed bf16 (hi(a) | hi(b) << 16) and the packed bf16 of the remainders
__device__ __forceinline__ void tw_split_pair(float a, float b, unsigned& hi, unsigned& lo)
{
    const tw_f32x2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, tw_bf16x2));
    const tw_f32x2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, tw_bf16x2));
}

__global__ void __launch_bounds__(SM_THREADS, 2)
thin_wgrad_x3_kernel(const float* __restrict__ dense, int Cd, const float* __restrict__ s0, const float* __restrict__ s1,
                     int Cs, int sign, float* __restrict__ part, int B, int X, int Y, int Z, int ntx, int nty, int ntz,
                     int ntiles, const TwPerm perm)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sD = smem;
    char* sS = smem + TW_D_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, li = lane & 31;
    const int cb = blockIdx.y;
    const int KG = 27 * Cs;
    // lane-constant fragment offsets
    int aoff[2], boff[3];
#pragma unroll
    for (int m = 0; m < 2; ++m) aoff[m] = (m * 32 + li) * TW_DSTR;
    const int dsw = tw_dswz(li);                    // = tw_dswz(32 + li)
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const int g = perm.g[n * 32 + li] & 127;
        int o = 0;
        if (g < KG) {
            const int tap = g / Cs, c = g - tap * Cs;
            const int ox = sign * (tap / 9 - 1), oy = sign * ((tap / 3) % 3 - 1), oz = sign * (tap % 3 - 1);
            o = (c * 3 + oz + 1) * TW_SPL + ((ox + 1) * TW_HY + (oy + 1)) * 16;
        }
        boff[n] = o;
    }
    // staging roles: dense = (z-row tid >> 4, channel quad tid & 15) x 8 z;  S = (haloed (x, y) row, channel) x 10 z
    const int drow = tid >> 4, dc4 = tid & 15;
    const int nS = TW_HX * TW_HY * Cs;
    const bool s_on = tid < nS;
    const int sc = s_on ? tid / (TW_HX * TW_HY) : 0;
    const int shxy = s_on ? tid - sc * (TW_HX * TW_HY) : 0;
    const float* sbase = s1 ? (sc ? s1 : s0) : s0 + sc;
    const int sstr = s1 ? 1 : Cs;
    const size_t nvox = (size_t)X * Y * Z;

    tw_u32x4 dpf[8];
    float spf[10];
    unsigned dmask = 0, smask = 0;
    auto issue = [&](int tile) {
        int t = tile;
        const int tzi = t % ntz; t /= ntz;
        const int tyi = t % nty; t /= nty;
        const int txi = t % ntx;
        const int b = t / ntx;
        const int x0 = txi * TW_TX, y0 = tyi * TW_TY, z0 = tzi * TW_TZ;
        {
            const int gx = x0 + (drow >> 3), gy = y0 + (drow & 7);
            const bool okr = gx < X && gy < Y;
            const float* p = dense + ((size_t)b * nvox + ((size_t)min(gx, X - 1) * Y + min(gy, Y - 1)) * Z) * Cd + cb * 64 +
                             dc4 * 4;
            unsigned mk = 0;
#pragma unroll
            for (int z = 0; z < 8; ++z) {
                const int gz = z0 + z;
                mk |= (okr && gz < Z) ? (1u << z) : 0u;
                dpf[z] = *reinterpret_cast<const tw_u32x4*>(p + (size_t)min(gz, Z - 1) * Cd);
            }
            dmask = mk;
        }
        if (s_on) {
            const int gx = x0 + shxy / TW_HY - 1, gy = y0 + shxy % TW_HY - 1;
            const bool okr = gx >= 0 && gx < X && gy >= 0 && gy < Y;
            const float* p = sbase + ((size_t)b * nvox + ((size_t)min(max(gx, 0), X - 1) * Y + min(max(gy, 0), Y - 1)) * Z) * sstr;
            unsigned mk = 0;
#pragma unroll
            for (int hz = 0; hz < 10; ++hz) {
                const int gz = z0 + hz - 1;
                mk |= (okr && gz >= 0 && gz < Z) ? (1u << hz) : 0u;
                spf[hz] = p[(size_t)min(max(gz, 0), Z - 1) * sstr];
            }
            smask = mk;
        }
    };

    f32x16 acc[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // passes of the persistent grid; every XCD walks a contiguous run of each pass's tiles (xcd_tile: neighbouring tiles share halo
    // rows, and with the plain walk they sit on different XCDs)
    int pass = 0, tile = xcd_tile((int)blockIdx.x, (int)gridDim.x, 0, ntiles), tile_next = -1;
    if (tile >= 0) issue(tile);
    for (; tile >= 0; tile = tile_next, ++pass) {
        tile_next = xcd_tile((int)blockIdx.x, (int)gridDim.x, pass + 1, ntiles);
        // fp32 -> packed (hi, lo) cells in registers
        tw_u32x4 dh[4], dl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = (dmask >> (2 * q)) & 1 ? __uint_as_float(dpf[2 * q][j]) : 0.f;
                const float b2 = (dmask >> (2 * q + 1)) & 1 ? __uint_as_float(dpf[2 * q + 1][j]) : 0.f;
                unsigned hi, lo;
                tw_split_pair(a, b2, hi, lo);
                dh[j][q] = hi; dl[j][q] = lo;
            }
        unsigned se_h[5], se_l[5], so_h[4], so_l[4];   // pairs (2q, 2q+1) and (2q+1, 2q+2) of the 10 z values
        {
            float v[10];
#pragma unroll
            for (int hz = 0; hz < 10; ++hz) v[hz] = (smask >> hz) & 1 ? spf[hz] : 0.f;
#pragma unroll
            for (int q = 0; q < 5; ++q) tw_split_pair(v[2 * q], v[2 * q + 1], se_h[q], se_l[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                so_h[q] = (se_h[q] >> 16) | (se_h[q + 1] << 16);
                so_l[q] = (se_l[q] >> 16) | (se_l[q + 1] << 16);
            }
        }
        __syncthreads();   // every wave has finished the previous tile's fragments
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            char* o = sD + (dc4 * 4 + j) * TW_DSTR + ((drow ^ tw_dswz(dc4 * 4 + j)) << 4);
            *reinterpret_cast<tw_u32x4*>(o) = dh[j];
            *reinterpret_cast<tw_u32x4*>(o + TW_DHL) = dl[j];
        }
        if (s_on) {
            char* o = sS + sc * 3 * TW_SPL + shxy * 16;
            const tw_u32x4 w0h = {se_h[0], se_h[1], se_h[2], se_h[3]}, w0l = {se_l[0], se_l[1], se_l[2], se_l[3]};
            const tw_u32x4 w1h = {so_h[0], so_h[1], so_h[2], so_h[3]}, w1l = {so_l[0], so_l[1], so_l[2], so_l[3]};
            const tw_u32x4 w2h = {se_h[1], se_h[2], se_h[3], se_h[4]}, w2l = {se_l[1], se_l[2], se_l[3], se_l[4]};
            *reinterpret_cast<tw_u32x4*>(o) = w0h;               *reinterpret_cast<tw_u32x4*>(o + TW_SHL) = w0l;
            *reinterpret_cast<tw_u32x4*>(o + TW_SPL) = w1h;      *reinterpret_cast<tw_u32x4*>(o + TW_SPL + TW_SHL) = w1l;
            *reinterpret_cast<tw_u32x4*>(o + 2 * TW_SPL) = w2h;  *reinterpret_cast<tw_u32x4*>(o + 2 * TW_SPL + TW_SHL) = w2l;
        }
        __syncthreads();
        if (tile_next >= 0) issue(tile_next);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = wave * 4 + e * 2 + h;
            const char* ap = sD + ((row ^ dsw) << 4);
            const char* bp = sS + ((row >> 3) * TW_HY + (row & 7)) * 16;
            bf16x8_t ah[2], al[2], bh[3], bl[3];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8_t*>(ap + aoff[m]);
                al[m] = *reinterpret_cast<const bf16x8_t*>(ap + aoff[m] + TW_DHL);
            }
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                bh[n] = *reinterpret_cast<const bf16x8_t*>(bp + boff[n]);
                bl[n] = *reinterpret_cast<const bf16x8_t*>(bp + boff[n] + TW_SHL);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                }
        }
    }
    // ordered sum of the four wave partials through LDS, one partial per workgroup
    float* red = reinterpret_cast<float*>(smem);   // [64][96]
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int d = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        float* o = red + d * 96 + n * 32 + li;
                        *o = (w == 0) ? acc[m][n][r] : *o + acc[m][n][r];
                    }
        }
    }
    __syncthreads();
    float* o = part + ((size_t)blockIdx.x * gridDim.y + cb) * (64 * 96);
    for (int i = tid; i < 64 * 96; i += SM_THREADS) {     // back to natural column order (padding columns are dropped)
        const int d = i / 96, j = i - d * 96;
        const int g = perm.g[j];
        if (g < 128) o[d * 96 + g] = red[i];
    }
}

// Deal the KG = 27 Cs real columns over the six 16-lane groups of the S fragment reads (positions n * 32 + {0-3, 12-15, 20-27}
// and n * 32 + {4-11, 16-19, 28-31}) so that the 16-B cells of a group lie in distinct 4-bank slots: slot = (byte offset / 16)
// mod 16 of the lane-constant part of the address (the row-dependent part is the same for the whole group).
static TwPerm tw_make_perm(int Cs, int sign)
{
    const int KG = 27 * Cs;
    static const int G1[16] = {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27};
    static const int G2[16] = {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31};
    int slot[96], cnt[6] = {0, 0, 0, 0, 0, 0}, member[6][16];
    bool used[6][16] = {};
    for (int g = 0; g < KG; ++g) {
        const int tap = g / Cs, c = g - tap * Cs;
        const int ox = sign * (tap / 9 - 1), oy = sign * ((tap / 3) % 3 - 1), oz = sign * (tap % 3 - 1);
        slot[g] = (((c * 3 + oz + 1) * TW_SPL + ((ox + 1) * TW_HY + (oy + 1)) * 16) / 16) & 15;
    }
    bool ok = true;
    for (int sl = 0; sl < 16 && ok; ++sl)
        for (int g = 0; g < KG && ok; ++g) {
            if (slot[g] != sl) continue;
            int best = -1;
            for (int k = 0; k < 6; ++k)
                if (!used[k][sl] && cnt[k] < 16 && (best < 0 || cnt[k] < cnt[best])) best = k;
            if (best < 0) { ok = false; break; }
            used[best][sl] = true;
            member[best][cnt[best]++] = g;
        }
    TwPerm p;
    if (!ok) {   // (never for Cs = 2, 3; any order is correct, only slower)
        for (int j = 0; j < 96; ++j) p.g[j] = (unsigned char)(j < KG ? j : 128);
        return p;
    }
    for (int k = 0; k < 6; ++k) {
        const int n = k >> 1;
        const int* lanes = (k & 1) ? G2 : G1;
        for (int i = 0; i < 16; ++i) {
            const int j = n * 32 + lanes[i];
            if (i < cnt[k]) p.g[j] = (unsigned char)member[k][i];
            else p.g[j] = (unsigned char)(128 | (cnt[k] ? member[k][0] : 0));   // padding: broadcast of the group's first cell
        }
    }
    return p;
}

static int launch_thin_wgrad_x3(const float* dense, int Cd, const float* s0, const float* s1, int Cs, int sign, float* dw,
                                void* ws, int B, int X, int Y, int Z, int mode, int accumulate, void* stream)
{
    const int ntx = (X + TW_TX - 1) / TW_TX, nty = (Y + TW_TY - 1) / TW_TY, ntz = (Z + TW_TZ - 1) / TW_TZ;
    const int ntiles = B * ntx * nty * ntz;
    const int ncb = Cd / 64;
    int gx = 512 / ncb;
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    const TwPerm perm = tw_make_perm(Cs, sign);
    hipLaunchKernelGGL(thin_wgrad_x3_kernel, dim3(gx, ncb), dim3(SM_THREADS), TW_LDS, as_stream(stream), dense, Cd, s0, s1,
                       Cs, sign, (float*)ws, B, X, Y, Z, ntx, nty, ntz, ntiles, perm);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(smallch_wgrad_reduce_kernel, dim3(Cd * 27 * Cs), dim3(64), 0, as_stream(stream), (const float*)ws, dw,
                       gx, ncb, Cd, Cs, mode, accumulate);
    return check_launch();
}

// ------------------------------------------------------------------------- //
// Flow-head data gradient on the bf16 cores (fp32x3)                          //
// ------------------------------------------------------------------------- //
// dX^T[ci][v] = sum_k W^T[ci][k] * G^T[k][v]: channels are the MFMA rows, so a lane ends up with 4 consecutive channels
// of one voxel (16-B stores / mask loads instead of 4-B ones).  The contraction index is ordered (dx, dy | dz, co):
// for a fixed in-plane shift the 9 (dz, co) values of a voxel are 9 consecutive floats of dY, stored per haloed voxel
// as one 16-element bf16 cell (9 used; the weight image is zero elsewhere), i.e. a k-step of 16 = one (dx, dy) and a
// fragment = one ds_read_b128; K = 9 x 16 = 144 for 81 real taps.  LDS: cells [hi|lo][half][(x, y) row 40][z 8] x 16 B,
// weights [hi|lo][k-step 9][half][ci 64] x 16 B, raw dY halo [40][10][3] fp32.  Persistent over 2 x 8 x 8 tiles (one
// 32-voxel column block per wave); the next tile's dY halo is prefetched into registers, the LeakyReLU mask of the
// current tile is loaded under the MFMAs; the bias-gradient column sums stay in registers until the end.
constexpr int FD_CELLS = TW_HX * TW_HY * 8;        // 320
constexpr int FD_G_PLANE = FD_CELLS * 16;          // per (hi|lo, half)
constexpr int FD_G_BYTES = 4 * FD_G_PLANE;         // 20,480
constexpr int FD_W_PLANE = 9 * 2 * 64 * 16;        // per hi|lo: 18,432
constexpr int FD_W_BYTES = 2 * FD_W_PLANE;         // 36,864
constexpr int FD_RAW = TW_HX * TW_HY * 30;         // 1200 floats
constexpr int FD_LDS = FD_G_BYTES + FD_W_BYTES + FD_RAW * 4;   // 62,144
constexpr int FD_RAW_IT = (FD_RAW + SM_THREADS - 1) / SM_THREADS;   // 5

__global__ void __launch_bounds__(SM_THREADS, 2)
flow_dgrad_x3_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int B, int X, int Y,
                     int Z, int Cin, int ntx, int nty, int ntz, int ntiles, const float* __restrict__ ymask, float alpha,
                     double* __restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sG = smem;
    char* sW = smem + FD_G_BYTES;
    float* sRaw = reinterpret_cast<float*>(smem + FD_G_BYTES + FD_W_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, li = lane & 31;
    const int cb = blockIdx.y;
    const size_t nvox = (size_t)X * Y * Z;

    for (int i = tid; i < 9 * 64 * 16; i += SM_THREADS) {   // weight image, once per workgroup
        const int idx = i & 15, ci = (i >> 4) & 63, ks = i >> 10;
        float f = 0.f;
        if (idx < 9) {
            const int tap = (2 - ks / 3) * 9 + (2 - ks % 3) * 3 + (2 - idx / 3);
            f = w[((size_t)tap * Cin + cb * 64 + ci) * 3 + idx % 3];
        }
        unsigned short hi, lo;
        split_bf16(f, hi, lo);
        const int off = ((ks * 2 + (idx >> 3)) * 64 + ci) * 16 + (idx & 7) * 2;
        *reinterpret_cast<unsigned short*>(sW + off) = hi;
        *reinterpret_cast<unsigned short*>(sW + FD_W_PLANE + off) = lo;
    }

    float rpf[FD_RAW_IT];
    unsigned rmask = 0;
    auto issue = [&](int tile) {
        int t = tile;
        const int tzi = t % ntz; t /= ntz;
        const int tyi = t % nty; t /= nty;
        const int txi = t % ntx;
        const int b = t / ntx;
        const int x0 = txi * TW_TX, y0 = tyi * TW_TY, z0 = tzi * TW_TZ;
        unsigned mk = 0;
#pragma unroll
        for (int it = 0; it < FD_RAW_IT; ++it) {
            const int i = tid + it * SM_THREADS;
            const int ic = i < FD_RAW ? i : 0;
            const int hxy = ic / 30, r = ic - hxy * 30;
            const int gx = x0 + hxy / TW_HY - 1, gy = y0 + hxy % TW_HY - 1, gz = z0 + r / 3 - 1;
            const bool ok = i < FD_RAW && gx >= 0 && gx < X && gy >= 0 && gy < Y && gz >= 0 && gz < Z;
            mk |= ok ? (1u << it) : 0u;
            rpf[it] = dy[((size_t)b * nvox + ((size_t)min(max(gx, 0), X - 1) * Y + min(max(gy, 0), Y - 1)) * Z +
                          min(max(gz, 0), Z - 1)) * 3 + r % 3];
        }
        rmask = mk;
    };

    // this lane's voxel inside a tile and its fragment bases
    const int zrow = wave * 4 + (li >> 3);
    const int vx = zrow >> 3, vy = zrow & 7, vz = li & 7;
    const char* gbase = sG + ((vx * TW_HY + vy) * 8 + vz) * 16 + h * FD_G_PLANE;
    const char* wbase = sW + (h * 64 + li) * 16;

    float4 cs[2][4];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) cs[n][q] = make_float4(0.f, 0.f, 0.f, 0.f);

    int pass = 0, tile = xcd_tile((int)blockIdx.x, (int)gridDim.x, 0, ntiles), tile_next = -1;     // (see thin_wgrad_x3_kernel)
    if (tile >= 0) issue(tile);
    for (; tile >= 0; tile = tile_next, ++pass) {
        tile_next = xcd_tile((int)blockIdx.x, (int)gridDim.x, pass + 1, ntiles);
        int t = tile;
        const int tzi = t % ntz; t /= ntz;
        const int tyi = t % nty; t /= nty;
        const int txi = t % ntx;
        const int b = t / ntx;
        const int gx = txi * TW_TX + vx, gy = tyi * TW_TY + vy, gz = tzi * TW_TZ + vz;
        const bool vok = gx < X && gy < Y && gz < Z;
        const size_t e = vok ? ((size_t)b * nvox + ((size_t)gx * Y + gy) * Z + gz) * Cin + cb * 64 + 4 * h : 0;
#pragma unroll
        for (int it = 0; it < FD_RAW_IT; ++it) {
            const int i = tid + it * SM_THREADS;
            if (i < FD_RAW) sRaw[i] = (rmask >> it) & 1 ? rpf[it] : 0.f;
        }
        __syncthreads();   // raw halo complete; every wave is past the previous tile's fragments
        for (int c = tid; c < FD_CELLS; c += SM_THREADS) {
            const float* r = sRaw + (c >> 3) * 30 + (c & 7) * 3;
            unsigned hp[5], lp[5];
#pragma unroll
            for (int q = 0; q < 4; ++q) tw_split_pair(r[2 * q], r[2 * q + 1], hp[q], lp[q]);
            tw_split_pair(r[8], 0.f, hp[4], lp[4]);
            const tw_u32x4 h0 = {hp[0], hp[1], hp[2], hp[3]}, l0 = {lp[0], lp[1], lp[2], lp[3]};
            const tw_u32x4 h1 = {hp[4], 0u, 0u, 0u}, l1 = {lp[4], 0u, 0u, 0u};
            char* o = sG + c * 16;
            *reinterpret_cast<tw_u32x4*>(o) = h0;
            *reinterpret_cast<tw_u32x4*>(o + FD_G_PLANE) = h1;
            *reinterpret_cast<tw_u32x4*>(o + 2 * FD_G_PLANE) = l0;
            *reinterpret_cast<tw_u32x4*>(o + 3 * FD_G_PLANE) = l1;
        }
        __syncthreads();
        if (tile_next >= 0) issue(tile_next);
        float4 mk4[2][4];
        if (ymask) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) mk4[n][q] = *reinterpret_cast<const float4*>(ymask + e + n * 32 + 8 * q);
        }
        f32x16 acc[2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) {
            const char* gp = gbase + ((ks / 3) * TW_HY + ks % 3) * 8 * 16;
            const bf16x8_t g_hi = *reinterpret_cast<const bf16x8_t*>(gp);
            const bf16x8_t g_lo = *reinterpret_cast<const bf16x8_t*>(gp + 2 * FD_G_PLANE);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const char* wp = wbase + (ks * 2 * 64 + n * 32) * 16;
                const bf16x8_t w_hi = *reinterpret_cast<const bf16x8_t*>(wp);
                const bf16x8_t w_lo = *reinterpret_cast<const bf16x8_t*>(wp + FD_W_PLANE);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_lo, g_hi, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi, g_lo, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi, g_hi, acc[n], 0, 0, 0);
            }
        }
        if (vok) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4(acc[n][4 * q], acc[n][4 * q + 1], acc[n][4 * q + 2], acc[n][4 * q + 3]);
                    if (ymask) {   // LeakyReLU backward of the layer that made the flow head's input, and its bias gradient
                        const float4 m = mk4[n][q];
                        v.x *= m.x < 0.f ? alpha : 1.f; v.y *= m.y < 0.f ? alpha : 1.f;
                        v.z *= m.z < 0.f ? alpha : 1.f; v.w *= m.w < 0.f ? alpha : 1.f;
                        cs[n][q].x += v.x; cs[n][q].y += v.y; cs[n][q].z += v.z; cs[n][q].w += v.w;
                    }
                    *reinterpret_cast<float4*>(dx + e + n * 32 + 8 * q) = v;
                }
        }
    }
    if (ymask) {   // column sums: [wave][lane][n][q][4] through LDS, one double per channel and workgroup
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);   // 4 x 64 x 32 floats = 32 KB
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(red + ((wave * 64 + lane) * 8 + n * 4 + q) * 4) = cs[n][q];
        __syncthreads();
        if (tid < 64) {
            const int n = tid >> 5, q = (tid >> 3) & 3, hh = (tid >> 2) & 1, k = tid & 3;   // channel = n*32 + 8q + 4h + k
            double s = 0.0;
            for (int wv = 0; wv < 4; ++wv)
                for (int l = 0; l < 32; ++l) s += (double)red[((wv * 64 + hh * 32 + l) * 8 + n * 4 + q) * 4 + k];
            part[(size_t)blockIdx.x * Cin + cb * 64 + tid] = s;
        }
    }
}

static int launch_flow_dgrad_x3(const float* dy, const float* w, float* dx, int B, int X, int Y, int Z, int Cin,
                                const float* ymask, float alpha, float* dbias, void* ws, int accumulate, void* stream)
{
    const int ntx = (X + TW_TX - 1) / TW_TX, nty = (Y + TW_TY - 1) / TW_TY, ntz = (Z + TW_TZ - 1) / TW_TZ;
    const int64_t nt = (int64_t)B * ntx * nty * ntz;
    if (nt > 0x7fffffff) return MMR_EINVAL;
    const int ncb = Cin / 64;
    int gx = 512 / ncb;
    if (gx > nt) gx = (int)nt;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(flow_dgrad_x3_kernel, dim3(gx, ncb), dim3(SM_THREADS), FD_LDS, as_stream(stream), dy, w, dx, B, X, Y, Z,
                       Cin, ntx, nty, ntz, (int)nt, ymask, alpha, (double*)ws);
    if (ymask)
        hipLaunchKernelGGL(bias_final_kernel, dim3(Cin), dim3(64), 0, as_stream(stream), (const double*)ws, dbias, Cin, gx,
                           accumulate);
    return check_launch();
}

// flow-head dgrad: dX[v][ci] = sum_g G[v][g] * Wr[g][ci], g = tap*3 + co, G[v][g] = dY[v - off(tap)][co]
// (M = 256 voxels per block, N = 64 input channels per blockIdx.y, K = 81 padded to 82)
__global__ void __launch_bounds__(SM_THREADS, 2)
flow_dgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int B, int X,
                       int Y, int Z, int Cin, int ntx, int nty, int ntz, const float* __restrict__ ymask, float alpha,
                       double* __restrict__ part)
{
    __shared__ float sS[W_HROWS * 3];
    __shared__ float sW[82 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int cb = blockIdx.y;
    int bid = blockIdx.x;
    const int tzi = bid % ntz; bid /= ntz;
    const int tyi = bid % nty; bid /= nty;
    const int txi = bid % ntx;
    const int b = bid / ntx;
    const int x0 = txi * W_TX, y0 = tyi * W_TY, z0 = tzi * W_TZ;
    const size_t nvox = (size_t)X * Y * Z;
    for (int i = tid; i < W_HROWS; i += SM_THREADS) {
        const int hx = i / (W_HY * W_HZ), hy = (i / W_HZ) % W_HY, hz = i % W_HZ;
        const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
        const bool ok = gx >= 0 && gx < X && gy >= 0 && gy < Y && gz >= 0 && gz < Z;
        const size_t o = ((size_t)b * nvox + ((size_t)gx * Y + gy) * Z + gz) * 3;
        sS[i * 3] = ok ? dy[o] : 0.f;
        sS[i * 3 + 1] = ok ? dy[o + 1] : 0.f;
        sS[i * 3 + 2] = ok ? dy[o + 2] : 0.f;
    }
    for (int i = tid; i < 82 * 64; i += SM_THREADS) {
        const int g = i >> 6, ci = i & 63;
        sW[i] = (g < 81) ? w[((size_t)(g / 3) * Cin + cb * 64 + ci) * 3 + (g % 3)] : 0.f;
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    int hrow[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int v = wave * 64 + m * 32 + (lane & 31);
        hrow[m] = ((((v >> 6) + 1) * W_HY + ((v >> 3) & 7) + 1) * W_HZ + (v & 7) + 1) * 3;
    }
#pragma unroll
    for (int kk = 0; kk < 41; ++kk) {
        const int g0 = 2 * kk, g1 = 2 * kk + 1;  // compile-time; lane half h picks one
        const int t0 = g0 / 3, t1 = (g1 < 81) ? g1 / 3 : 0;
        const int o0 = -(((t0 / 9 - 1) * (W_HY * W_HZ) + ((t0 / 3) % 3 - 1) * W_HZ + (t0 % 3 - 1)) * 3) + g0 % 3;
        const int o1 = (g1 < 81) ? -(((t1 / 9 - 1) * (W_HY * W_HZ) + ((t1 / 3) % 3 - 1) * W_HZ + (t1 % 3 - 1)) * 3) + g1 % 3 : 0;
        const int off = h ? o1 : o0;
        const float a0 = sS[hrow[0] + off], a1 = sS[hrow[1] + off];
        const float b0 = sW[(g0 + h) * 64 + (lane & 31)], b1 = sW[(g0 + h) * 64 + 32 + (lane & 31)];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    float cs0 = 0.f, cs1 = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = wave * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int gx = x0 + (v >> 6), gy = y0 + ((v >> 3) & 7), gz = z0 + (v & 7);
            if (gx < X && gy < Y && gz < Z) {
                const size_t e = ((size_t)b * nvox + ((size_t)gx * Y + gy) * Z + gz) * Cin + cb * 64 + (lane & 31);
                float v0 = acc[m][0][r], v1 = acc[m][1][r];
                if (ymask) {  // LeakyReLU backward of the layer that made the flow head's input, and its bias gradient
                    if (ymask[e] < 0.f) v0 *= alpha;
                    if (ymask[e + 32] < 0.f) v1 *= alpha;
                    cs0 += v0;
                    cs1 += v1;
                }
                dx[e] = v0;
                dx[e + 32] = v1;
            }
        }
    if (ymask) {
        cs0 += __shfl_xor(cs0, 32);
        cs1 += __shfl_xor(cs1, 32);
        __syncthreads();  // every wave is past its sS reads
        if (h == 0) {
            sS[wave * 64 + (lane & 31)] = cs0;
            sS[wave * 64 + 32 + (lane & 31)] = cs1;
        }
        __syncthreads();
        if (tid < 64)
            part[(size_t)blockIdx.x * Cin + cb * 64 + tid] =
                (double)sS[tid] + (double)sS[64 + tid] + (double)sS[128 + tid] + (double)sS[192 + tid];
    }
}

// ------------------------------------------------------------------------- //
// Adam (Keras: w -= lr_t * m / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t))
// ------------------------------------------------------------------------- //
__global__ void __launch_bounds__(TB)
adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            int64_t n, float lr_t, float b1, float b2, float eps, float gscale)
{
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        w[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

inline int rblocks(int64_t n, int per = TB * 8, int cap = 1024)
{
    int64_t g = (n + per - 1) / per;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace mmr

using namespace mmr;

// ---- Dice from labels ---------------------------------------------------- //
extern "C" int64_t mmr_dice_labels_ws_bytes(int B, int64_t nvox, int L)
{
    if (B < 1 || nvox < 1 || L < 1 || L > 64) return MMR_EINVAL;
    return (int64_t)B * rblocks(nvox, TB * 16) * L * 2 * sizeof(double);
}

static int dice_labels_fwd_impl(const uint8_t* lab1, const uint8_t* lab2, const float* flow, float* loss,
                                float* top_bot, void* ws, int B, int X, int Y, int Z, int L, int zeropad, int dice_mode,
                                void* stream)
{
    if (!lab1 || !lab2 || !flow || !loss || !top_bot || !ws || B < 1 || X < 1 || Y < 1 || Z < 1 || L < 1 || L > 64)
        return MMR_EINVAL;
    if (dice_mode != MMR_DICE_DIVIDE_NO_NAN && dice_mode != MMR_DICE_MAX_EPS) return MMR_EINVAL;
    const int nblk = rblocks((int64_t)X * Y * Z, TB * 16);
    hipLaunchKernelGGL(dice_labels_partial_kernel, dim3(nblk, B), dim3(TB), 2 * L * TB * sizeof(float),
                       as_stream(stream), lab1, lab2, flow, (double*)ws, X, Y, Z, L, nblk, zeropad);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(dice_labels_sum_kernel, dim3(B * L), dim3(64), 0, as_stream(stream), (const double*)ws, top_bot, L,
                       nblk);
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(dice_labels_final_kernel, dim3(1), dim3(TB), 0, as_stream(stream), (const float*)top_bot, loss, B,
                       L, zeropad, dice_mode);
    return check_launch();
}

extern "C" int mmr_dice_labels_fwd(const uint8_t* lab1, const uint8_t* lab2, const float* flow, float* loss,
                                   float* top_bot, void* ws, int B, int X, int Y, int Z, int L, int dice_mode,
                                   void* stream)
{
    return dice_labels_fwd_impl(lab1, lab2, flow, loss, top_bot, ws, B, X, Y, Z, L, 0, dice_mode, stream);
}

extern "C" int mmr_dice_labels_zeropad_fwd(const uint8_t* lab1, const uint8_t* lab2, const float* flow, float* loss,
                                           float* top_bot, void* ws, int B, int X, int Y, int Z, int L, int dice_mode,
                                           void* stream)
{
    if (L < 2) return MMR_EINVAL;
    return dice_labels_fwd_impl(lab1, lab2, flow, loss, top_bot, ws, B, X, Y, Z, L, 1, dice_mode, stream);
}

static int dice_labels_bwd_impl(const uint8_t* lab1, const uint8_t* lab2, const float* flow, const float* top_bot,
                                float* dflow, int B, int X, int Y, int Z, int L, float scale, int accumulate, int zeropad,
                                int dice_mode, void* stream)
{
    if (!lab1 || !lab2 || !flow || !top_bot || !dflow || B < 1 || X < 1 || Y < 1 || Z < 1 || L < 1 || L > 64)
        return MMR_EINVAL;
    if (dice_mode != MMR_DICE_DIVIDE_NO_NAN && dice_mode != MMR_DICE_MAX_EPS) return MMR_EINVAL;
    hipLaunchKernelGGL(dice_labels_bwd_kernel, dim3(rblocks((int64_t)X * Y * Z, TB * 4, 2048), B), dim3(TB), 0,
                       as_stream(stream), lab1, lab2, flow, top_bot, dflow, B, X, Y, Z, L, scale, accumulate, zeropad,
                       dice_mode);
    return check_launch();
}

extern "C" int mmr_dice_labels_bwd(const uint8_t* lab1, const uint8_t* lab2, const float* flow, const float* top_bot,
                                   float* dflow, int B, int X, int Y, int Z, int L, float scale, int accumulate,
                                   int dice_mode, void* stream)
{
    return dice_labels_bwd_impl(lab1, lab2, flow, top_bot, dflow, B, X, Y, Z, L, scale, accumulate, 0, dice_mode, stream);
}

extern "C" int mmr_dice_labels_zeropad_bwd(const uint8_t* lab1, const uint8_t* lab2, const float* flow,
                                           const float* top_bot, float* dflow, int B, int X, int Y, int Z, int L,
                                           float scale, int accumulate, int dice_mode, void* stream)
{
    return dice_labels_bwd_impl(lab1, lab2, flow, top_bot, dflow, B, X, Y, Z, L, scale, accumulate, 1, dice_mode, stream);
}

extern "C" int mmr_grad_l2_bwd_f32(const float* flow, float* dflow, int B, int X, int Y, int Z, int C, float loss_mult,
                                   float scale, int accumulate, void* stream)
{
    if (!flow || !dflow || B < 1 || X < 2 || Y < 2 || Z < 2 || C < 1) return MMR_EINVAL;
    const double k = (double)scale * loss_mult / 3.0 * 2.0;
    const float cx = (float)(k / ((double)(X - 1) * Y * Z * C));
    const float cy = (float)(k / ((double)X * (Y - 1) * Z * C));
    const float cz = (float)(k / ((double)X * Y * (Z - 1) * C));
    hipLaunchKernelGGL(grad_l2_bwd_kernel, dim3(stream_grid((int64_t)B * X * Y * 64, TB)), dim3(TB), 0,
                       as_stream(stream), flow, dflow, B, X, Y, Z, C, cx, cy, cz, accumulate);
    return check_launch();
}

extern "C" int mmr_resize_trilinear_bwd_f32(const float* dout, float* din, int B, int X, int Y, int Z, int C, int Xo,
                                            int Yo, int Zo, float mul, int grid_mode, float zoom, void* stream)
{
    if (!dout || !din || B < 1 || X < 1 || Y < 1 || Z < 1 || C < 1 || Xo < 1 || Yo < 1 || Zo < 1) return MMR_EINVAL;
    hipStream_t st = as_stream(stream);
    float stx, sty, stz;
    if (resize_steps(X, Y, Z, Xo, Yo, Zo, grid_mode, zoom, stx, sty, stz)) return MMR_EINVAL;
    hipLaunchKernelGGL(resize_bwd_kernel, dim3(stream_grid((int64_t)B * X * Y * Z * C, TB)), dim3(TB), 0, st, dout,
                       din, B, X, Y, Z, C, Xo, Yo, Zo, stx, sty, stz, mul);
    return check_launch();
}

// The same adjoint as three per-axis passes (resize_bwd_axis_kernel) through two intermediates in `ws`:
// [B, X, Yo, Zo, C] after the x pass, [B, X, Y, Zo, C] after the y pass.
extern "C" int64_t mmr_resize_trilinear_bwd_ws_bytes(int B, int X, int Y, int Z, int C, int Xo, int Yo, int Zo)
{
    if (B < 1 || X < 1 || Y < 1 || Z < 1 || C < 1 || Xo < 1 || Yo < 1 || Zo < 1) return MMR_EINVAL;
    return ((int64_t)B * X * Yo * Zo * C + (int64_t)B * X * Y * Zo * C) * (int64_t)sizeof(float);
}

extern "C" int mmr_resize_trilinear_bwd_ws_f32(const float* dout, float* din, void* ws, int B, int X, int Y, int Z, int C, int Xo,
                                               int Yo, int Zo, float mul, int grid_mode, float zoom, void* stream)
{
    if (!dout || !din || !ws || B < 1 || X < 1 || Y < 1 || Z < 1 || C < 1 || Xo < 1 || Yo < 1 || Zo < 1) return MMR_EINVAL;
    hipStream_t st = as_stream(stream);
    float stx, sty, stz;
    if (resize_steps(X, Y, Z, Xo, Yo, Zo, grid_mode, zoom, stx, sty, stz)) return MMR_EINVAL;
    float* t1 = static_cast<float*>(ws);
    float* t2 = t1 + (int64_t)B * X * Yo * Zo * C;
    launch_resize_bwd_axis(dout, t1, (int64_t)B, X, Xo, (int64_t)Yo * Zo * C, stx, 1.0f, st);
    launch_resize_bwd_axis((const float*)t1, t2, (int64_t)B * X, Y, Yo, (int64_t)Zo * C, sty, 1.0f, st);
    launch_resize_bwd_axis((const float*)t2, din, (int64_t)B * X * Y, Z, Zo, (int64_t)C, stz, mul, st);
    return check_launch();
}

// da, db zeroed by the caller.
static int launch_compose_bwd(const float* a, const float* b, const float* dout, float* da, float* db, int B, int X, int Y,
                              int Z, float s, hipStream_t st)
{
    const int ntx = (X + CB_TX - 1) / CB_TX, nty = (Y + CB_TY - 1) / CB_TY, ntz = (Z + CB_TZ - 1) / CB_TZ;
    const int64_t nt = (int64_t)B * ntx * nty * ntz;
    if (nt > 0x7fffffff)   // the one-atomic-per-corner kernel only when the tile count overflows the grid
        hipLaunchKernelGGL(compose_bwd_kernel, dim3(stream_grid((int64_t)B * X * Y * Z, TB)), dim3(TB), 0, st, a, b, dout,
                           da, db, B, X, Y, Z, s);
    else
        hipLaunchKernelGGL(compose_bwd_tiled_kernel, dim3((unsigned)nt), dim3(TB), 0, st, a, b, dout, da, db, B, X, Y, Z, s,
                           ntx, nty, ntz);
    return check_launch();
}

extern "C" int mmr_compose_bwd_f32(const float* a, const float* b, const float* dout, float* da, float* db, int B,
                                   int X, int Y, int Z, void* stream)
{
    if (!a || !b || !dout || !da || !db || B < 1 || X < 1 || Y < 1 || Z < 1) return MMR_EINVAL;
    hipStream_t st = as_stream(stream);
    const size_t bytes = (size_t)B * X * Y * Z * 3 * sizeof(float);
    if (hipMemsetAsync(da, 0, bytes, st) != hipSuccess) return MMR_EHIP;
    if (db != da && hipMemsetAsync(db, 0, bytes, st) != hipSuccess) return MMR_EHIP;
    return launch_compose_bwd(a, b, dout, da, db, B, X, Y, Z, 1.0f, st);
}

// steps: [nsteps][B,X,Y,Z,3] = the input of every squaring step as saved by mmr_vecint_save_f32
extern "C" int mmr_vecint_save_f32(const float* vel, float* steps, float* out, int B, int X, int Y, int Z, int nsteps,
                                   void* stream);

extern "C" int mmr_vecint_bwd_f32(const float* vel, const float* steps, const float* dout, float* dvel, float* tmp,
                                  int B, int X, int Y, int Z, int nsteps, void* stream)
{
    if (!vel || !dout || !dvel || !tmp || B < 1 || X < 1 || Y < 1 || Z < 1 || nsteps < 0 || nsteps > 30) return MMR_EINVAL;
    if (nsteps > 1 && !steps) return MMR_EINVAL;
    hipStream_t st = as_stream(stream);
    const int64_t nel = (int64_t)B * X * Y * Z * 3;
    if (nsteps == 0) {
        if (hipMemcpyAsync(dvel, dout, nel * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return MMR_EHIP;
        return MMR_OK;
    }
    // step k (k = nsteps-1 .. 0) consumed input steps[k] (k >= 1) or vel scaled by 2^-n (k = 0)
    const float* g = dout;
    float* bufs[2] = {dvel, tmp};
    int cur = (nsteps % 2 == 1) ? 0 : 1;  // so that the last written buffer is dvel
    for (int k = nsteps - 1; k >= 0; --k) {
        float* dst = bufs[cur];
        if (hipMemsetAsync(dst, 0, nel * sizeof(float), st) != hipSuccess) return MMR_EHIP;
        const float* in = (k == 0) ? vel : steps + (int64_t)(k - 1) * nel;
        const float s = (k == 0) ? 1.0f / (float)(1 << nsteps) : 1.0f;
        int rc = launch_compose_bwd(in, in, g, dst, dst, B, X, Y, Z, s, st);
        if (rc) return rc;
        g = dst;
        cur ^= 1;
    }
    return MMR_OK;
}

extern "C" int mmr_warp3d_bwd_flow_f32(const float* vol, const float* flow, const float* dout, float* dflow, int B,
                                       int X, int Y, int Z, int C, void* stream)
{
    if (!vol || !flow || !dout || !dflow || B < 1 || X < 1 || Y < 1 || Z < 1 || C < 1) return MMR_EINVAL;
    hipLaunchKernelGGL(warp_bwd_flow_kernel, dim3(stream_grid((int64_t)B * X * Y * Z, TB)), dim3(TB), 0,
                       as_stream(stream), vol, flow, dout, dflow, B, X, Y, Z, C);
    return check_launch();
}

extern "C" int mmr_warp3d_bwd_vol_f32(const float* flow, const float* dout, float* dvol, int B, int X, int Y, int Z,
                                      int C, void* stream)
{
    if (!flow || !dout || !dvol || B < 1 || X < 1 || Y < 1 || Z < 1 || C < 1) return MMR_EINVAL;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(dvol, 0, (size_t)B * X * Y * Z * C * sizeof(float), st) != hipSuccess) return MMR_EHIP;
    hipLaunchKernelGGL(warp_bwd_vol_kernel, dim3(stream_grid((int64_t)B * X * Y * Z * C, TB)), dim3(TB), 0, st, flow,
                       dout, dvol, B, X, Y, Z, C);
    return check_launch();
}

extern "C" int mmr_conv3d_k3_wgrad_f32x1(const float* in0, int C0, int up0, const float* in1, int C1, const float* dz,
                                         float* dw, void* ws, int B, int X, int Y, int Z, int Cout, int accumulate,
                                         void* stream);

// ---- conv backward plumbing ----------------------------------------------- //
extern "C" int64_t mmr_leaky_bwd_ws_bytes(int64_t nvox, int C)
{
    if (nvox < 1 || C < 1 || C > 256) return MMR_EINVAL;
    return (int64_t)rblocks(nvox * C) * C * sizeof(double);
}

extern "C" int mmr_leaky_bwd_bias_f32(const float* y, const float* dy, float* dz, float* dbias, void* ws, int64_t nvox,
                                      int C, int leaky, float alpha, int accumulate, void* stream)
{
    if (!dy || !dz || !dbias || !ws || nvox < 1 || C < 1 || C > 256 || (leaky && !y)) return MMR_EINVAL;
    const int nblk = rblocks(nvox * C);
    hipLaunchKernelGGL(leaky_bwd_bias_kernel, dim3(nblk), dim3(TB), TB * sizeof(double), as_stream(stream), y, dy, dz,
                       (double*)ws, nvox, C, alpha, leaky, nblk);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(bias_final_kernel, dim3(C), dim3(64), 0, as_stream(stream), (const double*)ws, dbias,
                       C, nblk, accumulate);
    return check_launch();
}

namespace {
constexpr int V4_BLOCKS = 2048;
inline bool v4_ok(int c) { return c > 0 && c % 4 == 0 && c <= 4 * TB; }
}  // namespace

extern "C" int64_t mmr_upcat_bwd_masked_ws_bytes(int C0, int C1)
{
    return (int64_t)V4_BLOCKS * (C0 + C1) * (int64_t)sizeof(double);
}

// Split the gradient of concat([up2(in0) | in0, in1]) and push each part through the LeakyReLU backward of the layer
// that produced it: d_in0 = pool-sum(dcat[..., :C0]) * L'(y0), d_in1 (+)= dcat[..., C0:] * L'(y1); the bias gradients
// of those layers (+)= the column sums.  y0 / y1 (activated outputs, same shapes as d_in0 / d_in1) may be null.
extern "C" int mmr_upcat_bwd_masked_f32(const float* dcat, float* d_in0, float* d_in1, int B, int X, int Y, int Z, int C0,
                                        int C1, int up0, int accumulate_in1, const float* y0, const float* y1, float alpha,
                                        float* dbias0, int acc_b0, float* dbias1, int acc_b1, void* ws, void* stream)
{
    if (!dcat || !d_in0 || B < 1 || X < 1 || Y < 1 || Z < 1 || C0 < 1 || C1 < 0 || (C1 > 0 && !d_in1)) return MMR_EINVAL;
    if (up0 && ((X | Y | Z) & 1)) return MMR_EINVAL;
    if ((y0 && !dbias0) || (y1 && (!dbias1 || C1 == 0)) || ((y0 || y1) && !ws)) return MMR_EINVAL;
    if (!v4_ok(C0) || (C1 > 0 && !v4_ok(C1))) return MMR_EUNSUPPORTED;
    const int64_t nvox = (int64_t)B * X * Y * Z;
    int64_t g = (nvox * ((C0 + C1) / 4) + TB * 4 - 1) / (TB * 4);
    const int nblk = (int)(g < 1 ? 1 : (g > V4_BLOCKS ? V4_BLOCKS : g));
    double* part0 = (double*)ws;
    double* part1 = part0 + (int64_t)V4_BLOCKS * C0;
    hipLaunchKernelGGL(upcat_bwd_v4_kernel, dim3(nblk), dim3(TB), 0, as_stream(stream), dcat, d_in0, d_in1, B, X, Y, Z, C0,
                       C1, up0, accumulate_in1, y0, y1, alpha, part0, part1);
    if (y0)
        hipLaunchKernelGGL(bias_final_kernel, dim3(C0), dim3(64), 0, as_stream(stream), (const double*)part0, dbias0, C0,
                           nblk, acc_b0);
    if (y1)
        hipLaunchKernelGGL(bias_final_kernel, dim3(C1), dim3(64), 0, as_stream(stream), (const double*)part1, dbias1, C1,
                           nblk, acc_b1);
    return check_launch();
}

extern "C" int mmr_upcat_bwd_f32(const float* dcat, float* d_in0, float* d_in1, int B, int X, int Y, int Z, int C0,
                                 int C1, int up0, int accumulate_in1, void* stream)
{
    if (!dcat || !d_in0 || B < 1 || X < 1 || Y < 1 || Z < 1 || C0 < 1 || C1 < 0 || (C1 > 0 && !d_in1)) return MMR_EINVAL;
    if (up0 && ((X | Y | Z) & 1)) return MMR_EINVAL;
    if (v4_ok(C0) && (C1 == 0 || v4_ok(C1)))
        return mmr_upcat_bwd_masked_f32(dcat, d_in0, d_in1, B, X, Y, Z, C0, C1, up0, accumulate_in1, nullptr, nullptr, 0.f,
                                        nullptr, 0, nullptr, 0, nullptr, stream);
    hipLaunchKernelGGL(upcat_bwd_kernel, dim3(stream_grid((int64_t)B * X * Y * Z * (C0 + C1), TB)), dim3(TB), 0,
                       as_stream(stream), dcat, d_in0, d_in1, B, X, Y, Z, C0, C1, up0, accumulate_in1);
    return check_launch();
}

extern "C" int64_t mmr_maxpool3d2_bwd_masked_ws_bytes(int C) { return (int64_t)V4_BLOCKS * C * (int64_t)sizeof(double); }

// MaxPooling3D(2) backward; with ymask semantics (`masked`): x is the activated output of a LeakyReLU layer and dx its
// PRE-activation gradient: the routed value is multiplied by L'(x at the maximum) and dbias (+)= its column sums.
extern "C" int mmr_maxpool3d2_bwd_masked_f32(const float* x, const float* dpool, float* dx, int B, int X, int Y, int Z,
                                             int C, int accumulate, int masked, float alpha, float* dbias, int acc_b,
                                             void* ws, void* stream)
{
    if (!x || !dpool || !dx || B < 1 || X < 2 || Y < 2 || Z < 2 || C < 1 || ((X | Y | Z) & 1)) return MMR_EINVAL;
    if (masked && (!dbias || !ws)) return MMR_EINVAL;
    if (!v4_ok(C)) return MMR_EUNSUPPORTED;
    const int64_t nh = (int64_t)B * (X / 2) * (Y / 2) * (Z / 2);
    int64_t g = (nh * (C / 4) + TB - 1) / TB;
    const int nblk = (int)(g < 1 ? 1 : (g > V4_BLOCKS ? V4_BLOCKS : g));
    hipLaunchKernelGGL(maxpool_bwd_v4_kernel, dim3(nblk), dim3(TB), 0, as_stream(stream), x, dpool, dx, B, X, Y, Z, C,
                       accumulate, masked, alpha, (double*)ws);
    if (masked)
        hipLaunchKernelGGL(bias_final_kernel, dim3(C), dim3(64), 0, as_stream(stream), (const double*)ws, dbias, C, nblk,
                           acc_b);
    return check_launch();
}

extern "C" int mmr_maxpool3d2_bwd_f32(const float* x, const float* dpool, float* dx, int B, int X, int Y, int Z, int C,
                                      int accumulate, void* stream)
{
    if (!x || !dpool || !dx || B < 1 || X < 2 || Y < 2 || Z < 2 || C < 1 || ((X | Y | Z) & 1)) return MMR_EINVAL;
    if (v4_ok(C))
        return mmr_maxpool3d2_bwd_masked_f32(x, dpool, dx, B, X, Y, Z, C, accumulate, 0, 0.f, nullptr, 0, nullptr, stream);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(stream_grid((int64_t)B * (X / 2) * (Y / 2) * (Z / 2) * C, TB)),
                       dim3(TB), 0, as_stream(stream), x, dpool, dx, B, X, Y, Z, C, accumulate);
    return check_launch();
}

namespace {
inline void wgrad_geom(int B, int X, int Y, int Z, int Cin, int Cout, int& ntx, int& nty, int& ntz, int& ntiles,
                       int& nslices, int& ncob, int& gx)
{
    ntx = (X + W_TX - 1) / W_TX; nty = (Y + W_TY - 1) / W_TY; ntz = (Z + W_TZ - 1) / W_TZ;
    ntiles = B * ntx * nty * ntz;
    nslices = Cin / 32;
    ncob = (Cout + 63) / 64;
    gx = 256 / (nslices * ncob);
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
}
}  // namespace

extern "C" int64_t mmr_conv3d_k3_wgrad_ws_bytes(int B, int X, int Y, int Z, int Cin, int Cout)
{
    if (B < 1 || X < 1 || Y < 1 || Z < 1 || Cin < 32 || Cin % 32 || Cout < 1) return MMR_EINVAL;
    int ntx, nty, ntz, ntiles, nslices, ncob, gx;
    wgrad_geom(B, X, Y, Z, Cin, Cout, ntx, nty, ntz, ntiles, nslices, ncob, gx);
    int64_t bytes = (int64_t)gx * nslices * ncob * 27 * 32 * 64 * sizeof(float);
    const int64_t thin = (int64_t)512 * 4 * 64 * 96 * sizeof(float);  // smallch_wgrad_kernel partials
    if (Cout == 3 && thin > bytes) bytes = thin;
    return bytes;
}

// dW (Keras layout [27][C0+C1][Cout]) (+)= wgrad of conv(concat([up2(in0)|in0, in1])) given dZ [B,X,Y,Z,Cout]
static int wgrad_impl(const float* in0, int C0, int up0, const float* in1, int C1, const float* dz, float* dw, void* ws,
                      int B, int X, int Y, int Z, int Cout, int accumulate, int x3, void* stream, int cin_total = 0, int ci_off = 0)
{
    if (cin_total == 0) cin_total = C0 + C1;     // dw is the layer's whole kernel [27][cin_total][Cout]; this call fills rows
    if (ci_off < 0 || ci_off + C0 + C1 > cin_total) return MMR_EINVAL;   // ci_off .. ci_off + C0 + C1 of every tap
    if (!in0 || !dz || !dw || !ws || B < 1 || X < 1 || Y < 1 || Z < 1 || Cout < 1 || C0 < 32 || C0 % 32 || C1 < 0 ||
        C1 % 32 || (C1 > 0 && !in1))
        return MMR_EINVAL;
    if (up0 && ((X | Y | Z) & 1)) return MMR_EINVAL;
    if (Cout == 3 && C1 == 0 && !up0 && C0 % 64 == 0) {  // flow head: taps folded into the GEMM N axis (81 of 96 used)
        if (x3 == 1)
            return launch_thin_wgrad_x3(in0, C0, dz, nullptr, 3, -1, dw, ws, B, X, Y, Z, 0, accumulate, stream);
        const int ntx = (X + W_TX - 1) / W_TX, nty = (Y + W_TY - 1) / W_TY, ntz = (Z + W_TZ - 1) / W_TZ;
        const int ntiles = B * ntx * nty * ntz;
        const int ncb = C0 / 64;
        int gx = 512 / ncb;
        if (gx > ntiles) gx = ntiles;
        if (gx < 1) gx = 1;
        const size_t lds = (size_t)(W_HROWS * 3 + 256 * 64) * sizeof(float);
        static bool attr0 = false;
        if (!attr0) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(smallch_wgrad_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
            attr0 = true;
        }
        hipLaunchKernelGGL(smallch_wgrad_kernel, dim3(gx, ncb), dim3(SM_THREADS), lds, as_stream(stream), in0, C0, dz,
                           (const float*)nullptr, 3, -1, (float*)ws, B, X, Y, Z, ntx, nty, ntz, ntiles);
        int rc0 = check_launch();
        if (rc0) return rc0;
        hipLaunchKernelGGL(smallch_wgrad_reduce_kernel, dim3(C0 * 81), dim3(64), 0,
                           as_stream(stream), (const float*)ws, dw, gx * 4, ncb, C0, 3, 0, accumulate);
        return check_launch();
    }
    WgradParams p;
    p.in0 = in0; p.in1 = in1; p.dz = dz; p.slab = (float*)ws;
    p.B = B; p.X = X; p.Y = Y; p.Z = Z; p.C0 = C0; p.C1 = C1; p.up0 = up0; p.Cout = Cout;
    int nslices, ncob, gx;
    wgrad_geom(B, X, Y, Z, C0 + C1, Cout, p.ntx, p.nty, p.ntz, p.ntiles, nslices, ncob, gx);
    {
        const uint64_t nv = (uint64_t)B * X * Y * Z, lim = 0xF0000000ull - 64;
        const uint64_t b0 = (up0 ? nv / 8 : nv) * (uint64_t)C0 * 4, b1 = nv * (uint64_t)C1 * 4, bz = nv * (uint64_t)Cout * 4;
        const bool fits = b0 <= lim && b1 <= lim && bz <= lim;
        p.bytes_src[0] = fits ? (unsigned)b0 : 0;
        p.bytes_src[1] = fits ? (unsigned)b1 : 0;
        p.bytes_dz = fits ? (unsigned)bz : 0;
    }
    const bool usebuf = p.bytes_dz != 0;
    constexpr int LDS = W_A_BYTES + W_B_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
        attr_set = true;
    }
    if (x3) {
        constexpr int LDSX = WX_A_BYTES + WX_B_BYTES;
        static bool attr_x3 = false;
        if (!attr_x3) {
            const void* ks[] = {reinterpret_cast<const void*>(wgrad_x3_kernel<2, true>),
                                reinterpret_cast<const void*>(wgrad_x3_kernel<1, true>),
                                reinterpret_cast<const void*>(wgrad_x3_kernel<2, false>),
                                reinterpret_cast<const void*>(wgrad_x3_kernel<1, false>),
                                reinterpret_cast<const void*>(wgrad_x3_kernel<2, true, false, 3, true>),
                                reinterpret_cast<const void*>(wgrad_x3_kernel<2, false, false, 3, true>),
                                reinterpret_cast<const void*>(wgrad_x3_kernel<2, true, false, 3, true, true>),
                                reinterpret_cast<const void*>(wgrad_x3_kernel<2, false, false, 3, true, true>),
#ifdef MMR_DIAG
                                reinterpret_cast<const void*>(wgrad_x3_kernel<2, true, true, 3, true, true>),
#endif
            };
            for (size_t i = 0; i < sizeof(ks) / sizeof(ks[0]); ++i) {
                hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDSX);
                if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
            }
            attr_x3 = true;
        }
        const dim3 g3(gx, nslices, ncob), b3(W_THREADS);
        if (x3 == 1) {
            // Cout % 64 == 0: one branch-free dZ load path; usebuf: buffer loads with hardware bounds (tensors < 4 GB).
            // Both operand tiles of the next voxel tile are prefetched under the k-loop (PF = 3)
            const bool fullco = (Cout % 64) == 0;
#ifdef MMR_DIAG
            const bool stamp = g_diag_stamps != 0;   // mmr_debug_set_stamps(1), tools/wgrad_stamps.py
            if (fullco && usebuf && stamp) hipLaunchKernelGGL((wgrad_x3_kernel<2, true, true, 3, true, true>), g3, b3, LDSX, as_stream(stream), p);
            else
#endif
            if (Cout <= 32) hipLaunchKernelGGL((wgrad_x3_kernel<1, true>), g3, b3, LDSX, as_stream(stream), p);
            else if (fullco && usebuf) hipLaunchKernelGGL((wgrad_x3_kernel<2, true, false, 3, true, true>), g3, b3, LDSX, as_stream(stream), p);
            else if (fullco) hipLaunchKernelGGL((wgrad_x3_kernel<2, true, false, 3, true>), g3, b3, LDSX, as_stream(stream), p);
            else hipLaunchKernelGGL((wgrad_x3_kernel<2, true>), g3, b3, LDSX, as_stream(stream), p);
        } else {
            if (Cout <= 32) hipLaunchKernelGGL((wgrad_x3_kernel<1, false>), g3, b3, LDSX, as_stream(stream), p);
            else if ((Cout % 64) == 0 && usebuf) hipLaunchKernelGGL((wgrad_x3_kernel<2, false, false, 3, true, true>), g3, b3, LDSX, as_stream(stream), p);
            else if ((Cout % 64) == 0) hipLaunchKernelGGL((wgrad_x3_kernel<2, false, false, 3, true>), g3, b3, LDSX, as_stream(stream), p);
            else hipLaunchKernelGGL((wgrad_x3_kernel<2, false>), g3, b3, LDSX, as_stream(stream), p);
        }
    } else if (Cout <= 32)
        hipLaunchKernelGGL(wgrad_kernel<1>, dim3(gx, nslices, ncob), dim3(W_THREADS), LDS, as_stream(stream), p);
    else
        hipLaunchKernelGGL(wgrad_kernel<2>, dim3(gx, nslices, ncob), dim3(W_THREADS), LDS, as_stream(stream), p);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(stream_grid((int64_t)27 * (C0 + C1) * Cout, TB)), dim3(TB), 0,
                       as_stream(stream), (const float*)ws, dw, gx, nslices, ncob, C0 + C1, Cout, accumulate, cin_total, ci_off);
    return check_launch();
}

// ---- weight gradient of a decoder layer conv(concat([UpSampling3D(2)(x_low) | skip])), folded (see wgrad_x3_kernel FOLD) ---- //
namespace {
inline void wgrad_fold_geom(int B, int X2, int Y2, int Z2, int C0, int Cout, int& ntx, int& nty, int& ntz, int& ntiles, int& nslices,
                            int& ncob, int& gx)
{
    ntx = (X2 + W_TX - 1) / W_TX; nty = (Y2 + W_TY - 1) / W_TY; ntz = (Z2 + W_TZ - 1) / W_TZ;
    ntiles = B * ntx * nty * ntz;
    nslices = C0 / 32;
    ncob = Cout / 64;
    gx = 256 / (nslices * ncob * 2);   // two workgroups (four parity classes each) per (voxel-tile run, slice, cob)
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
}
}  // namespace

extern "C" int64_t mmr_conv3d_k3_wgrad_upfold_ws_bytes(int B, int X2, int Y2, int Z2, int C0, int C1, int Cout)
{
    if (B < 1 || X2 < 1 || Y2 < 1 || Z2 < 1 || C0 < 32 || C0 % 32 || C1 < 32 || C1 % 32 || Cout < 64 || Cout % 64) return MMR_EINVAL;
    int ntx, nty, ntz, ntiles, nslices, ncob, gx;
    wgrad_fold_geom(B, X2, Y2, Z2, C0, Cout, ntx, nty, ntz, ntiles, nslices, ncob, gx);
    const int64_t fold = (int64_t)gx * nslices * ncob * 8 * 27 * 2048 * sizeof(float);
    const int64_t skip = mmr_conv3d_k3_wgrad_ws_bytes(B, 2 * X2, 2 * Y2, 2 * Z2, C1, Cout);
    return fold > skip ? fold : skip;
}

// dw [27][C0 + C1][Cout] (+)= the layer's weight gradient given dz [B,2X2,2Y2,2Z2,Cout]: rows [0, C0) from the low-resolution
// x_low [B,X2,Y2,Z2,C0] through the folded correlation (64 class-tap products over N / 8 voxels instead of 27 over N), rows
// [C0, C0 + C1) from skip [B,2X2,2Y2,2Z2,C1] through the ordinary kernel.  x3mode 1 = fp32x3, 2 = bf16 hi products only.
extern "C" int mmr_conv3d_k3_wgrad_upfold(const float* x_low, int C0, const float* skip, int C1, const float* dz, float* dw,
                                          void* ws, int B, int X2, int Y2, int Z2, int Cout, int accumulate, int x3mode,
                                          void* stream)
{
    if (!x_low || !skip || !dz || !dw || !ws || (x3mode != 1 && x3mode != 2)) return MMR_EINVAL;
    if (mmr_conv3d_k3_wgrad_upfold_ws_bytes(B, X2, Y2, Z2, C0, C1, Cout) < 0) return MMR_EINVAL;
    const uint64_t nvl = (uint64_t)B * X2 * Y2 * Z2, lim = 0xF0000000ull - 64;
    if (nvl * C0 * 4 > lim || nvl * 8 * (uint64_t)Cout * 4 > lim) return MMR_EUNSUPPORTED;   // buffer-descriptor path only
    // skip half first (it may overwrite its rows), then the folded half into rows [0, C0)
    int rc = wgrad_impl(skip, C1, 0, nullptr, 0, dz, dw, ws, B, 2 * X2, 2 * Y2, 2 * Z2, Cout, accumulate, x3mode, stream, C0 + C1, C0);
    if (rc) return rc;
    WgradParams p;
    p.in0 = x_low; p.in1 = nullptr; p.dz = dz; p.slab = (float*)ws;
    p.B = B; p.X = X2; p.Y = Y2; p.Z = Z2; p.C0 = C0; p.C1 = 0; p.up0 = 0; p.Cout = Cout;
    int nslices, ncob, gx;
    wgrad_fold_geom(B, X2, Y2, Z2, C0, Cout, p.ntx, p.nty, p.ntz, p.ntiles, nslices, ncob, gx);
    p.bytes_src[0] = (unsigned)(nvl * C0 * 4);
    p.bytes_src[1] = 0;
    p.bytes_dz = (unsigned)(nvl * 8 * (uint64_t)Cout * 4);
    constexpr int LDSX = WX_A_BYTES + WX_B_BYTES;
    static bool attr = false;
    if (!attr) {
        const void* ks[] = {reinterpret_cast<const void*>(wgrad_x3_kernel<2, true, false, 3, true, true, true>),
                            reinterpret_cast<const void*>(wgrad_x3_kernel<2, false, false, 3, true, true, true>)};
        for (size_t i = 0; i < 2; ++i) {
            hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDSX);
            if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
        }
        attr = true;
    }
    const dim3 g3(gx, nslices, ncob * 2), b3(W_THREADS);
    if (x3mode == 1) hipLaunchKernelGGL((wgrad_x3_kernel<2, true, false, 3, true, true, true>), g3, b3, LDSX, as_stream(stream), p);
    else hipLaunchKernelGGL((wgrad_x3_kernel<2, false, false, 3, true, true, true>), g3, b3, LDSX, as_stream(stream), p);
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_fold_reduce_kernel, dim3(stream_grid((int64_t)27 * C0 * Cout, TB)), dim3(TB), 0, as_stream(stream),
                       (const float*)ws, dw, gx, nslices, ncob, C0, Cout, accumulate, C0 + C1);
    return check_launch();
}

#ifdef MMR_DIAG
// Diagnostic: copy out and clear the wgrad cycle stamps (tools/wgrad_stamps.py); not part of mmr.h.
extern "C" int mmr_debug_wgrad_stamps(unsigned long long* out64)
{
    unsigned long long z[64] = {0};
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_wgrad_stamp), sizeof(z)) != hipSuccess) return MMR_EHIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_stamp), z, sizeof(z)) != hipSuccess) return MMR_EHIP;
    return MMR_OK;
}
#endif

extern "C" int mmr_conv3d_k3_wgrad_f32(const float* in0, int C0, int up0, const float* in1, int C1, const float* dz,
                                       float* dw, void* ws, int B, int X, int Y, int Z, int Cout, int accumulate,
                                       void* stream)
{
    return wgrad_impl(in0, C0, up0, in1, C1, dz, dw, ws, B, X, Y, Z, Cout, accumulate, 0, stream);
}

// same contract; products are bf16 hi/lo splits (three bf16 MFMAs each), fp32-grade accuracy
extern "C" int mmr_conv3d_k3_wgrad_f32x3(const float* in0, int C0, int up0, const float* in1, int C1, const float* dz,
                                         float* dw, void* ws, int B, int X, int Y, int Z, int Cout, int accumulate,
                                         void* stream)
{
    return wgrad_impl(in0, C0, up0, in1, C1, dz, dw, ws, B, X, Y, Z, Cout, accumulate, 1, stream);
}

extern "C" int mmr_conv3d_k3_wgrad_f32x1(const float* in0, int C0, int up0, const float* in1, int C1, const float* dz,
                                         float* dw, void* ws, int B, int X, int Y, int Z, int Cout, int accumulate,
                                         void* stream)
{
    return wgrad_impl(in0, C0, up0, in1, C1, dz, dw, ws, B, X, Y, Z, Cout, accumulate, 2, stream);
}

extern "C" int64_t mmr_conv3d_k3_cin2_wgrad_ws_bytes(int Cout)
{
    if (Cout < 1) return MMR_EINVAL;
    const int64_t a = (int64_t)512 * 54 * Cout * sizeof(float);
    const int64_t b = (int64_t)512 * 4 * 64 * 96 * sizeof(float);
    return a > b ? a : b;
}

static int cin2_wgrad_impl(const float* src, const float* trg, const float* dz, float* dw, void* ws, int B, int X, int Y,
                           int Z, int Cout, int accumulate, int x3, void* stream)
{
    if (!src || !trg || !dz || !dw || !ws || B < 1 || X < 1 || Y < 1 || Z < 1) return MMR_EINVAL;
    if (Cout % 64 == 0 && x3)
        return launch_thin_wgrad_x3(dz, Cout, src, trg, 2, +1, dw, ws, B, X, Y, Z, 1, accumulate, stream);
    if (Cout % 64 == 0) {  // fp32 matrix cores: T[co][tap*2 + ci] = sum_v dZ[v][co] * img_ci[v + off(tap)]
        const int ntx = (X + W_TX - 1) / W_TX, nty = (Y + W_TY - 1) / W_TY, ntz = (Z + W_TZ - 1) / W_TZ;
        const int ntiles = B * ntx * nty * ntz;
        const int ncb = Cout / 64;
        int gx = 512 / ncb;
        if (gx > ntiles) gx = ntiles;
        if (gx < 1) gx = 1;
        const size_t lds = (size_t)(W_HROWS * 3 + 256 * 64) * sizeof(float);
        static bool attr0 = false;
        if (!attr0) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(smallch_wgrad_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
            attr0 = true;
        }
        hipLaunchKernelGGL(smallch_wgrad_kernel, dim3(gx, ncb), dim3(SM_THREADS), lds, as_stream(stream), dz, Cout, src, trg,
                           2, +1, (float*)ws, B, X, Y, Z, ntx, nty, ntz, ntiles);
        int rc0 = check_launch();
        if (rc0) return rc0;
        hipLaunchKernelGGL(smallch_wgrad_reduce_kernel, dim3(Cout * 54), dim3(64), 0,
                           as_stream(stream), (const float*)ws, dw, gx * 4, ncb, Cout, 2, 1, accumulate);
        return check_launch();
    }
    if (!(Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256)) return MMR_EUNSUPPORTED;
    const int ntx = (X + G_TX - 1) / G_TX, nty = (Y + G_TY - 1) / G_TY, ntz = (Z + G_TZ - 1) / G_TZ;
    const int ntiles = B * ntx * nty * ntz;
    const int nblk = ntiles < 512 ? ntiles : 512;
    hipLaunchKernelGGL(wgrad_cin2_kernel, dim3(nblk), dim3(TB), 0, as_stream(stream), src, trg, dz, (float*)ws, B, X, Y,
                       Z, Cout, ntx, nty, ntz, ntiles);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(stream_grid(54 * Cout, TB)), dim3(TB), 0, as_stream(stream),
                       (const float*)ws, dw, (int64_t)54 * Cout, nblk, accumulate);
    return check_launch();
}

extern "C" int mmr_conv3d_k3_cin2_wgrad_f32(const float* src, const float* trg, const float* dz, float* dw, void* ws,
                                            int B, int X, int Y, int Z, int Cout, int accumulate, void* stream)
{
    return cin2_wgrad_impl(src, trg, dz, dw, ws, B, X, Y, Z, Cout, accumulate, 0, stream);
}

// same contract; products as bf16 hi/lo splits on the bf16 MFMA (Cout % 64 == 0; other widths use the exact path)
extern "C" int mmr_conv3d_k3_cin2_wgrad_f32x3(const float* src, const float* trg, const float* dz, float* dw, void* ws,
                                              int B, int X, int Y, int Z, int Cout, int accumulate, void* stream)
{
    return cin2_wgrad_impl(src, trg, dz, dw, ws, B, X, Y, Z, Cout, accumulate, 1, stream);
}

extern "C" int mmr_conv3d_k3_cout3_dgrad_f32(const float* dy, const float* w_keras, float* dx, int B, int X, int Y,
                                             int Z, int Cin, void* stream)
{
    if (!dy || !w_keras || !dx || B < 1 || X < 1 || Y < 1 || Z < 1 || Cin < 1 || Cin > 512) return MMR_EINVAL;
    if (Cin % 64 == 0) {  // fp32 matrix cores, K = 27 taps x 3 folded
        const int ntx = (X + W_TX - 1) / W_TX, nty = (Y + W_TY - 1) / W_TY, ntz = (Z + W_TZ - 1) / W_TZ;
        const int64_t nblk = (int64_t)B * ntx * nty * ntz;
        if (nblk > 0x7fffffff) return MMR_EINVAL;
        hipLaunchKernelGGL(flow_dgrad_mfma_kernel, dim3((unsigned)nblk, Cin / 64), dim3(SM_THREADS), 0, as_stream(stream),
                           dy, w_keras, dx, B, X, Y, Z, Cin, ntx, nty, ntz, (const float*)nullptr, 0.f, (double*)nullptr);
        return check_launch();
    }
    if (Cin % 4 == 0) {
        const int64_t nblk = ((int64_t)B * X * Y * Z + TB - 1) / TB;
        if (nblk > 0x7fffffff) return MMR_EINVAL;
        hipLaunchKernelGGL(dgrad_cout3_kernel, dim3((unsigned)nblk), dim3(TB), 0, as_stream(stream), dy, w_keras, dx, B,
                           X, Y, Z, Cin);
        return check_launch();
    }
    const size_t lds = (size_t)27 * Cin * 3 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_cout3_generic_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 27 * 512 * 3 * sizeof(float));
        if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL(dgrad_cout3_generic_kernel, dim3(stream_grid((int64_t)B * X * Y * Z * Cin, TB)), dim3(TB), lds,
                       as_stream(stream), dy, w_keras, dx, B, X, Y, Z, Cin);
    return check_launch();
}

// same contract as mmr_conv3d_k3_cout3_dgrad_f32, bf16 hi/lo split products (Cin % 64 == 0, else MMR_EUNSUPPORTED)
extern "C" int mmr_conv3d_k3_cout3_dgrad_f32x3(const float* dy, const float* w_keras, float* dx, int B, int X, int Y,
                                               int Z, int Cin, void* stream)
{
    if (!dy || !w_keras || !dx || B < 1 || X < 1 || Y < 1 || Z < 1 || Cin < 1 || Cin > 512) return MMR_EINVAL;
    if (Cin % 64) return MMR_EUNSUPPORTED;
    return launch_flow_dgrad_x3(dy, w_keras, dx, B, X, Y, Z, Cin, nullptr, 0.f, nullptr, nullptr, 0, stream);
}

extern "C" int64_t mmr_conv3d_k3_cout3_dgrad_masked_ws_bytes(int B, int X, int Y, int Z, int Cin)
{
    // one double per (workgroup, channel); the exact path has one workgroup per 4 x 8 x 8 tile, the fp32x3 path at most 512
    const int64_t a = (int64_t)B * ((X + W_TX - 1) / W_TX) * ((Y + W_TY - 1) / W_TY) * ((Z + W_TZ - 1) / W_TZ);
    return (a > 512 ? a : 512) * Cin * (int64_t)sizeof(double);
}

// flow-head data gradient fused with the LeakyReLU backward + bias gradient of the layer feeding the flow head:
// dx = dgrad(dy) * (ymask < 0 ? alpha : 1); dbias (+)= sum_voxels dx.  Cin must be a multiple of 64 (MFMA kernel).
extern "C" int mmr_conv3d_k3_cout3_dgrad_masked_f32(const float* dy, const float* w_keras, float* dx, int B, int X, int Y,
                                                    int Z, int Cin, const float* ymask, float alpha, float* dbias, void* ws,
                                                    int accumulate, void* stream)
{
    if (!dy || !w_keras || !dx || !ymask || !dbias || !ws || B < 1 || X < 1 || Y < 1 || Z < 1 || Cin < 1 || Cin > 512)
        return MMR_EINVAL;
    if (Cin % 64) return MMR_EUNSUPPORTED;
    const int ntx = (X + W_TX - 1) / W_TX, nty = (Y + W_TY - 1) / W_TY, ntz = (Z + W_TZ - 1) / W_TZ;
    const int64_t nblk = (int64_t)B * ntx * nty * ntz;
    if (nblk > 0x7fffffff) return MMR_EINVAL;
    hipLaunchKernelGGL(flow_dgrad_mfma_kernel, dim3((unsigned)nblk, Cin / 64), dim3(SM_THREADS), 0, as_stream(stream),
                       dy, w_keras, dx, B, X, Y, Z, Cin, ntx, nty, ntz, ymask, alpha, (double*)ws);
    hipLaunchKernelGGL(bias_final_kernel, dim3(Cin), dim3(64), 0, as_stream(stream), (const double*)ws, dbias, Cin,
                       (int)nblk, accumulate);
    return check_launch();
}

extern "C" int mmr_conv3d_k3_cout3_dgrad_masked_f32x3(const float* dy, const float* w_keras, float* dx, int B, int X, int Y,
                                                      int Z, int Cin, const float* ymask, float alpha, float* dbias,
                                                      void* ws, int accumulate, void* stream)
{
    if (!dy || !w_keras || !dx || !ymask || !dbias || !ws || B < 1 || X < 1 || Y < 1 || Z < 1 || Cin < 1 || Cin > 512)
        return MMR_EINVAL;
    if (Cin % 64) return MMR_EUNSUPPORTED;
    return launch_flow_dgrad_x3(dy, w_keras, dx, B, X, Y, Z, Cin, ymask, alpha, dbias, ws, accumulate, stream);
}

extern "C" int mmr_adam_step_f32(float* w, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                 float beta2, float eps, int64_t step, float grad_scale, void* stream)
{
    if (!w || !g || !m || !v || n < 1 || step < 1) return MMR_EINVAL;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n, TB)), dim3(TB), 0, as_stream(stream), w, g, m, v, n, (float)lr_t,
                       beta1, beta2, eps, grad_scale);
    return check_launch();
}
