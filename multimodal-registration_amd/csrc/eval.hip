// Evaluation metrics of the reference's eval_reg_*.py scripts on device:
//   Jacobian determinant with 4th-order central differences (eval_reg_with_jacobian.py:62-78),
//   joint histogram for the normalized mutual information (eval_reg_with_mi.py:65-74, numpy histogramdd
//   semantics: right-closed last bin, edges supplied by the host), overlap sums on binary masks
//   (eval_reg_on_sc_seg.py:80-93).  fp64 like the reference's NumPy (get_fdata() is float64); all HBM-bound.
#include "common.hpp"

namespace mmr {

// ddf [X][Y][Z][3] (the reference's [X,Y,Z,1,3] with the singleton squeezed), det [X-4][Y-4][Z-4]
__global__ void __launch_bounds__(256)
jacobian_det_kernel(const double* __restrict__ ddf, double* __restrict__ det, int X, int Y, int Z)
{
    const int Xi = X - 4, Yi = Y - 4, Zi = Z - 4;
    const int64_t total = (int64_t)Xi * Yi * Zi;
    const int64_t sz = 3, sy = (int64_t)Z * 3, sx = (int64_t)Y * Z * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(i % Zi) + 2, y = (int)((i / Zi) % Yi) + 2, x = (int)(i / ((int64_t)Zi * Yi)) + 2;
        const double* p = ddf + x * sx + y * sy + z * sz;
        double J[3][3];  // J[channel][axis]
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            J[c][0] = (p[c - 2 * sx] - 8.0 * p[c - sx] + 8.0 * p[c + sx] - p[c + 2 * sx]) / 12.0;
            J[c][1] = (p[c - 2 * sy] - 8.0 * p[c - sy] + 8.0 * p[c + sy] - p[c + 2 * sy]) / 12.0;
            J[c][2] = (p[c - 2 * sz] - 8.0 * p[c - sz] + 8.0 * p[c + sz] - p[c + 2 * sz]) / 12.0;
        }
        J[0][0] += 1.0; J[1][1] += 1.0; J[2][2] += 1.0;
        det[i] = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                 J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    }
}

// hist[ba][bb] += 1; bin = searchsorted(edges, v, 'right') - 1, values equal to the last edge go to the last bin
__device__ __forceinline__ int hist_bin(const double* __restrict__ edges, int nb, double v)
{
    int lo = 0, hi = nb + 1;  // first index with edges[idx] > v
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (edges[mid] <= v) lo = mid + 1; else hi = mid;
    }
    int b = lo - 1;
    if (v == edges[nb]) b = nb - 1;
    return b;  // -1 or nb = outside (cannot happen when the edges span the data)
}

__global__ void __launch_bounds__(256)
joint_hist_kernel(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ ea,
                  const double* __restrict__ eb, unsigned long long* __restrict__ hist, int64_t n, int nb)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sa = reinterpret_cast<double*>(smem);
    double* sb = sa + nb + 1;
    for (int i = threadIdx.x; i <= nb; i += blockDim.x) { sa[i] = ea[i]; sb[i] = eb[i]; }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ia = hist_bin(sa, nb, a[i]), ib = hist_bin(sb, nb, b[i]);
        if (ia >= 0 && ia < nb && ib >= 0 && ib < nb) atomicAdd(hist + (int64_t)ia * nb + ib, 1ULL);
    }
}

// out[0..5] = sum m[f==1], sum m[f==0], count f==1, count f==0, sum m, n  (exact for 0/1 masks; double sums)
__global__ void __launch_bounds__(256)
overlap_sums_kernel(const double* __restrict__ f, const double* __restrict__ m, double* __restrict__ part, int64_t n)
{
    __shared__ double sh[4];
    double s[5] = {0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double fv = f[i], mv = m[i];
        if (fv == 1.0) { s[0] += mv; s[2] += 1.0; }
        if (fv == 0.0) { s[1] += mv; s[3] += 1.0; }
        s[4] += mv;
    }
    for (int k = 0; k < 5; ++k) {
        double v = wave_sum(s[k]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * 5 + k] = sh[0] + sh[1] + sh[2] + sh[3];
    }
}

__global__ void overlap_final_kernel(const double* __restrict__ part, double* __restrict__ out, int nblk, double n)
{
    const int k = threadIdx.x;
    if (k < 5) {
        double s = 0.0;
        for (int i = 0; i < nblk; ++i) s += part[(int64_t)i * 5 + k];
        out[k] = s;
    }
    if (k == 5) out[5] = n;
}

}  // namespace mmr

using namespace mmr;

extern "C" int mmr_jacobian_det_f64(const double* ddf, double* det, int X, int Y, int Z, void* stream)
{
    if (!ddf || !det || X < 5 || Y < 5 || Z < 5) return MMR_EINVAL;
    hipLaunchKernelGGL(jacobian_det_kernel, dim3(stream_grid((int64_t)(X - 4) * (Y - 4) * (Z - 4), 256)), dim3(256), 0,
                       as_stream(stream), ddf, det, X, Y, Z);
    return check_launch();
}

extern "C" int mmr_joint_hist_f64(const double* a, const double* b, const double* edges_a, const double* edges_b,
                                  unsigned long long* hist, int64_t n, int nbins, void* stream)
{
    if (!a || !b || !edges_a || !edges_b || !hist || n < 1 || nbins < 1 || nbins > 1024) return MMR_EINVAL;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(hist, 0, (size_t)nbins * nbins * sizeof(unsigned long long), st) != hipSuccess) return MMR_EHIP;
    hipLaunchKernelGGL(joint_hist_kernel, dim3(stream_grid(n, 256, 1024)), dim3(256), 2 * (nbins + 1) * sizeof(double), st, a,
                       b, edges_a, edges_b, hist, n, nbins);
    return check_launch();
}

extern "C" int64_t mmr_overlap_ws_bytes(void) { return (int64_t)1024 * 5 * sizeof(double); }

extern "C" int mmr_overlap_sums_f64(const double* fixed, const double* moved, double* out6, void* ws, int64_t n, void* stream)
{
    if (!fixed || !moved || !out6 || !ws || n < 1) return MMR_EINVAL;
    const int nblk = stream_grid(n, 256, 1024);
    hipLaunchKernelGGL(overlap_sums_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), fixed, moved, (double*)ws, n);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(overlap_final_kernel, dim3(1), dim3(64), 0, as_stream(stream), (const double*)ws, out6, nblk, (double)n);
    return check_launch();
}
