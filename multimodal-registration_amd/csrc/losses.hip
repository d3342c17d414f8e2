// Loss reductions (HBM-bound): Dice, Grad-l2, local NCC, bending energy.
// All reductions are two-stage and ordered (per-thread fp32 -> per-block
// double partials in a caller workspace -> one finalize block), so results
// are bitwise reproducible run to run -- no float atomics.
#include "common.hpp"

namespace mmr {

constexpr int RED_BLOCK = 256;

// block-wide sum of a double, result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sh /* >= 4 doubles */)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += sh[i];
    }
    return r;
}

// ------------------------------ Dice ------------------------------------ //
// y [B, nvox, L]; grid (nblk, B); threads >= T=(256/L)*L idle so that a
// thread's label is fixed while the block still reads contiguous runs.
__global__ void __launch_bounds__(RED_BLOCK)
dice_partial_kernel(const float* __restrict__ yt, const float* __restrict__ yp, double* __restrict__ part,
                    int64_t nvox, int L, int nblk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_top = reinterpret_cast<float*>(smem);
    float* s_bot = s_top + RED_BLOCK;
    const int T = (RED_BLOCK / L) * L;
    const int b = blockIdx.y;
    const int64_t n_el = nvox * L;
    const int64_t chunk = (((n_el + nblk - 1) / nblk + T - 1) / T) * T;  // multiple of T => multiple of L
    const int64_t lo = (int64_t)blockIdx.x * chunk;
    const int64_t hi = (lo + chunk < n_el) ? lo + chunk : n_el;
    float top = 0.f, bot = 0.f;
    if ((int)threadIdx.x < T) {
        const float* t = yt + (int64_t)b * n_el;
        const float* p = yp + (int64_t)b * n_el;
        for (int64_t e = lo + threadIdx.x; e < hi; e += T) {
            const float a = t[e], c = p[e];
            top += a * c;
            bot += a + c;
        }
    }
    s_top[threadIdx.x] = top;
    s_bot[threadIdx.x] = bot;
    __syncthreads();
    if ((int)threadIdx.x < L) {
        double st = 0.0, sb = 0.0;
        for (int k = threadIdx.x; k < T; k += L) {
            st += (double)s_top[k];
            sb += (double)s_bot[k];
        }
        double* o = part + (((int64_t)b * nblk + blockIdx.x) * L + threadIdx.x) * 2;
        o[0] = st;
        o[1] = sb;
    }
}

__global__ void __launch_bounds__(RED_BLOCK)
dice_final_kernel(const double* __restrict__ part, float* __restrict__ loss, float* __restrict__ top_bot,
                  int B, int L, int nblk)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < B * L; i += blockDim.x) {
        const int b = i / L, l = i % L;
        double st = 0.0, sb = 0.0;
        for (int k = 0; k < nblk; ++k) {
            const double* o = part + (((int64_t)b * nblk + k) * L + l) * 2;
            st += o[0];
            sb += o[1];
        }
        const float ft = (float)(2.0 * st), fb = (float)sb;
        if (top_bot) {
            top_bot[i * 2] = ft;
            top_bot[i * 2 + 1] = fb;
        }
        acc += (fb != 0.f) ? (double)(ft / fb) : 0.0;  // divide_no_nan
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = (float)(-s / (double)(B * L));
}

// ------------------------------ Grad-l2 --------------------------------- //
__global__ void __launch_bounds__(RED_BLOCK)
grad_l2_partial_kernel(const float* __restrict__ f, double* __restrict__ part, int X, int Y, int Z, int C, int nblk)
{
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const int64_t n_el = (int64_t)X * Y * Z * C;
    const float* p = f + (int64_t)b * n_el;
    const int64_t sz = C, sy = (int64_t)Z * C, sx = (int64_t)Y * Z * C;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (int64_t)nblk * blockDim.x) {
        const int64_t v = e / C;
        const int z = (int)(v % Z);
        const int y = (int)((v / Z) % Y);
        const int x = (int)(v / ((int64_t)Z * Y));
        const float c0 = p[e];
        if (x + 1 < X) { const float d = p[e + sx] - c0; ax += d * d; }
        if (y + 1 < Y) { const float d = p[e + sy] - c0; ay += d * d; }
        if (z + 1 < Z) { const float d = p[e + sz] - c0; az += d * d; }
    }
    const double rx = block_sum((double)ax, sh);
    const double ry = block_sum((double)ay, sh);
    const double rz = block_sum((double)az, sh);
    if (threadIdx.x == 0) {
        double* o = part + ((int64_t)b * nblk + blockIdx.x) * 3;
        o[0] = rx; o[1] = ry; o[2] = rz;
    }
}

__global__ void grad_l2_final_kernel(const double* __restrict__ part, float* __restrict__ out, int B, int X, int Y,
                                     int Z, int C, int nblk, float loss_mult)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s[3] = {0, 0, 0};
    for (int k = 0; k < nblk; ++k)
        for (int d = 0; d < 3; ++d) s[d] += part[((int64_t)b * nblk + k) * 3 + d];
    const double nx = (double)(X - 1) * Y * Z * C, ny = (double)X * (Y - 1) * Z * C, nz = (double)X * Y * (Z - 1) * C;
    const double m = (s[0] / nx + s[1] / ny + s[2] / nz) / 3.0;
    out[b] = (float)(m * (double)loss_mult);
}

// ------------------------------ local NCC ------------------------------- //
// Block = one (TY x TZ) column marched along an X segment.  Per plane: load
// the haloed tile of I and J, form I,J,I^2,J^2,IJ in LDS, box-sum along z
// then y (zero 'SAME' padding), keep a 9-deep register ring of plane sums per
// thread and sum the ring for the x box.  HBM traffic = inputs x halo factor.
template <int TY, int TZ, int WIN>
__global__ void __launch_bounds__(TY * TZ)
ncc_partial_kernel(const float* __restrict__ I, const float* __restrict__ J, double* __restrict__ part,
                   int X, int Y, int Z, int xseg, int nseg, float eps)
{
    constexpr int R = WIN / 2;           // 4
    constexpr int HY = TY + 2 * R, HZ = TZ + 2 * R;
    __shared__ float s_q[5][HY][HZ + 1];
    __shared__ float s_z[5][HY][TZ + 1];
    __shared__ double sh[(TY * TZ + 63) / 64 > 4 ? (TY * TZ + 63) / 64 : 4];
    const int tz = threadIdx.x % TZ, ty = threadIdx.x / TZ;
    const int b = blockIdx.z / nseg, seg = blockIdx.z % nseg;
    const int z0 = blockIdx.x * TZ, y0 = blockIdx.y * TY;
    const int x0 = seg * xseg;
    const int x1 = (x0 + xseg < X) ? x0 + xseg : X;
    const int64_t nvox = (int64_t)X * Y * Z;
    const float* Ib = I + (int64_t)b * nvox;
    const float* Jb = J + (int64_t)b * nvox;
    const float ws = (float)(WIN * WIN * WIN);
    float ring[WIN][5];
#pragma unroll
    for (int k = 0; k < WIN; ++k)
#pragma unroll
        for (int q = 0; q < 5; ++q) ring[k][q] = 0.f;
    float acc = 0.f;
    const bool own = (y0 + ty < Y) && (z0 + tz < Z);
    // planes xs = x0-R .. x1-1+R ; after pushing plane xs, voxel xo = xs-R is complete
    for (int xb = x0 - R; xb < x1 + R; xb += WIN) {
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const int xs = xb + k;
            if (xs < x1 + R) {  // uniform across the block
                const bool xin = (xs >= 0) && (xs < X);
                __syncthreads();
                for (int i = threadIdx.x; i < HY * HZ; i += TY * TZ) {
                    const int hz = i % HZ, hy = i / HZ;
                    const int yy = y0 + hy - R, zz = z0 + hz - R;
                    float a = 0.f, c = 0.f;
                    if (xin && yy >= 0 && yy < Y && zz >= 0 && zz < Z) {
                        const int64_t o = ((int64_t)xs * Y + yy) * Z + zz;
                        a = Ib[o];
                        c = Jb[o];
                    }
                    s_q[0][hy][hz] = a;
                    s_q[1][hy][hz] = c;
                    s_q[2][hy][hz] = a * a;
                    s_q[3][hy][hz] = c * c;
                    s_q[4][hy][hz] = a * c;
                }
                __syncthreads();
                for (int i = threadIdx.x; i < HY * TZ; i += TY * TZ) {
                    const int z = i % TZ, hy = i / TZ;
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        float s = 0.f;
#pragma unroll
                        for (int j = 0; j < WIN; ++j) s += s_q[q][hy][z + j];
                        s_z[q][hy][z] = s;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < WIN; ++j) s += s_z[q][ty + j][tz];
                    ring[k][q] = s;
                }
                const int xo = xs - R;
                if (own && xo >= x0 && xo < x1) {
                    float S[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        float s = 0.f;
#pragma unroll
                        for (int j = 0; j < WIN; ++j) s += ring[j][q];
                        S[q] = s;
                    }
                    const float uI = S[0] / ws, uJ = S[1] / ws;
                    const float cross = S[4] - uJ * S[0] - uI * S[1] + uI * uJ * ws;
                    const float Iv = S[2] - 2.f * uI * S[0] + uI * uI * ws;
                    const float Jv = S[3] - 2.f * uJ * S[1] + uJ * uJ * ws;
                    acc += cross * cross / (Iv * Jv + eps);
                }
            }
        }
    }
    const double r = block_sum((double)acc, sh);
    if (threadIdx.x == 0) {
        const int64_t nb = (int64_t)gridDim.x * gridDim.y * nseg;
        part[(int64_t)b * nb + ((int64_t)seg * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = r;
    }
}

__global__ void mean_final_kernel(const double* __restrict__ part, float* __restrict__ out, int B, int64_t nb,
                                  double denom, float sign)
{
    __shared__ double sh[4];
    const int b = blockIdx.x;
    double a = 0.0;
    for (int64_t k = threadIdx.x; k < nb; k += blockDim.x) a += part[(int64_t)b * nb + k];
    const double s = block_sum(a, sh);
    if (threadIdx.x == 0) out[b] = (float)((double)sign * s / denom);
}

// ------------------------------ bending --------------------------------- //
__global__ void __launch_bounds__(RED_BLOCK)
bending_partial_kernel(const float* __restrict__ u, double* __restrict__ part, int X, int Y, int Z, int nblk)
{
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const int64_t sz = 3, sy = (int64_t)Z * 3, sx = (int64_t)Y * Z * 3;
    const float* p = u + (int64_t)b * X * sx;
    const int Xi = X - 2, Yi = Y - 2, Zi = Z - 2;
    const int64_t n_el = (int64_t)Xi * Yi * Zi * 3;
    float acc = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (int64_t)nblk * blockDim.x) {
        const int c = (int)(e % 3);
        const int64_t v = e / 3;
        const int z = (int)(v % Zi) + 1;
        const int y = (int)((v / Zi) % Yi) + 1;
        const int x = (int)(v / ((int64_t)Zi * Yi)) + 1;
        const float* q = p + x * sx + y * sy + z * sz + c;
        const float c0 = q[0];
        const float dxx = q[sx] - 2.f * c0 + q[-sx];
        const float dyy = q[sy] - 2.f * c0 + q[-sy];
        const float dzz = q[sz] - 2.f * c0 + q[-sz];
        const float dxy = (q[sx + sy] - q[sx - sy] - q[-sx + sy] + q[-sx - sy]) * 0.25f;
        const float dxz = (q[sx + sz] - q[sx - sz] - q[-sx + sz] + q[-sx - sz]) * 0.25f;
        const float dyz = (q[sy + sz] - q[sy - sz] - q[-sy + sz] + q[-sy - sz]) * 0.25f;
        acc += dxx * dxx + dyy * dyy + dzz * dzz + 2.f * (dxy * dxy + dxz * dxz + dyz * dyz);
    }
    const double r = block_sum((double)acc, sh);
    if (threadIdx.x == 0) part[(int64_t)b * nblk + blockIdx.x] = r;
}

inline int red_blocks(int64_t n_el)
{
    int64_t g = (n_el + RED_BLOCK * 8 - 1) / (RED_BLOCK * 8);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace mmr

using namespace mmr;

extern "C" int64_t mmr_dice_ws_bytes(int B, int64_t nvox, int L)
{
    if (B < 1 || nvox < 1 || L < 1) return MMR_EINVAL;
    return (int64_t)B * red_blocks(nvox * L) * L * 2 * sizeof(double);
}

extern "C" int mmr_dice_fwd_f32(const float* y_true, const float* y_pred, float* loss_out, float* top_bot, void* ws,
                                int B, int64_t nvox, int L, void* stream)
{
    if (!y_true || !y_pred || !loss_out || !ws || B < 1 || nvox < 1 || L < 1 || L > RED_BLOCK) return MMR_EINVAL;
    const int nblk = red_blocks(nvox * L);
    hipLaunchKernelGGL(dice_partial_kernel, dim3(nblk, B), dim3(RED_BLOCK), 2 * RED_BLOCK * sizeof(float),
                       as_stream(stream), y_true, y_pred, (double*)ws, nvox, L, nblk);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(dice_final_kernel, dim3(1), dim3(RED_BLOCK), 0, as_stream(stream), (const double*)ws, loss_out,
                       top_bot, B, L, nblk);
    return check_launch();
}

extern "C" int64_t mmr_grad_l2_ws_bytes(int B, int X, int Y, int Z, int C)
{
    if (B < 1 || X < 1 || Y < 1 || Z < 1 || C < 1) return MMR_EINVAL;
    return (int64_t)B * red_blocks((int64_t)X * Y * Z * C) * 3 * sizeof(double);
}

extern "C" int mmr_grad_l2_fwd_f32(const float* flow, float* out, void* ws, int B, int X, int Y, int Z, int C,
                                   float loss_mult, void* stream)
{
    if (!flow || !out || !ws || B < 1 || X < 2 || Y < 2 || Z < 2 || C < 1) return MMR_EINVAL;
    const int nblk = red_blocks((int64_t)X * Y * Z * C);
    hipLaunchKernelGGL(grad_l2_partial_kernel, dim3(nblk, B), dim3(RED_BLOCK), 0, as_stream(stream), flow,
                       (double*)ws, X, Y, Z, C, nblk);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(grad_l2_final_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), (const double*)ws,
                       out, B, X, Y, Z, C, nblk, loss_mult);
    return check_launch();
}

namespace {
constexpr int NCC_TY = 8, NCC_TZ = 32, NCC_XSEG = 64;
inline void ncc_grid(int X, int Y, int Z, int& gz, int& gy, int& nseg)
{
    gz = (Z + NCC_TZ - 1) / NCC_TZ;
    gy = (Y + NCC_TY - 1) / NCC_TY;
    nseg = (X + NCC_XSEG - 1) / NCC_XSEG;
}
}  // namespace

extern "C" int64_t mmr_ncc_ws_bytes(int B, int X, int Y, int Z)
{
    if (B < 1 || X < 1 || Y < 1 || Z < 1) return MMR_EINVAL;
    int gz, gy, nseg;
    ncc_grid(X, Y, Z, gz, gy, nseg);
    return (int64_t)B * gz * gy * nseg * sizeof(double);
}

extern "C" int mmr_ncc_fwd_f32(const float* I, const float* J, float* out, void* ws, int B, int X, int Y, int Z,
                               int win, float eps, void* stream)
{
    if (!I || !J || !out || !ws || B < 1 || X < 1 || Y < 1 || Z < 1) return MMR_EINVAL;
    if (win != 9) return MMR_EUNSUPPORTED;
    int gz, gy, nseg;
    ncc_grid(X, Y, Z, gz, gy, nseg);
    if ((int64_t)B * nseg > 65535) return MMR_EINVAL;
    hipLaunchKernelGGL((ncc_partial_kernel<NCC_TY, NCC_TZ, 9>), dim3(gz, gy, B * nseg), dim3(NCC_TY * NCC_TZ), 0,
                       as_stream(stream), I, J, (double*)ws, X, Y, Z, NCC_XSEG, nseg, eps);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(mean_final_kernel, dim3(B), dim3(RED_BLOCK), 0, as_stream(stream), (const double*)ws, out, B,
                       (int64_t)gz * gy * nseg, (double)X * Y * Z, -1.0f);
    return check_launch();
}

extern "C" int64_t mmr_bending_ws_bytes(int B, int X, int Y, int Z)
{
    if (B < 1 || X < 3 || Y < 3 || Z < 3) return MMR_EINVAL;
    return (int64_t)B * red_blocks((int64_t)(X - 2) * (Y - 2) * (Z - 2) * 3) * sizeof(double);
}

extern "C" int mmr_bending_fwd_f32(const float* flow, float* out, void* ws, int B, int X, int Y, int Z, void* stream)
{
    if (!flow || !out || !ws || B < 1 || X < 3 || Y < 3 || Z < 3) return MMR_EINVAL;
    const int64_t n = (int64_t)(X - 2) * (Y - 2) * (Z - 2) * 3;
    const int nblk = red_blocks(n);
    hipLaunchKernelGGL(bending_partial_kernel, dim3(nblk, B), dim3(RED_BLOCK), 0, as_stream(stream), flow,
                       (double*)ws, X, Y, Z, nblk);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(mean_final_kernel, dim3(B), dim3(RED_BLOCK), 0, as_stream(stream), (const double*)ws, out, B,
                       (int64_t)nblk, (double)n, 1.0f);
    return check_launch();
}
