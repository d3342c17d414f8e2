// Loss reductions (HBM-bound): Dice, Grad-l2, local NCC, bending energy.
// All reductions are two-stage and ordered (per-thread fp32 -> per-block
// double partials in a caller workspace -> one finalize block), so results
// are bitwise reproducible run to run -- no float atomics.
#include <cstdlib>

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

// In-kernel finalize of a per-workgroup-partials reduction (no second launch): every workgroup publishes its double partial,
// then takes a ticket; the workgroup that draws the LAST ticket sums all partials in index order (the same order whichever
// workgroup that is -> bitwise reproducible, like mean_final_kernel) and writes out[b] = (+= when accumulate) sign * sum / denom.
// `ticket` must be ZERO when the kernel starts and is zero again when it ends (one ticket word per concurrently running call).
// Cross-XCD visibility WITHOUT fences: the partial is stored, the ticket bumped and the partials read back with agent-scope
// atomics (write-through / L2-bypassing accesses), ordered by an explicit vmcnt(0) between the store and the ticket.  A
// `__threadfence()` per workgroup measured 57 -> 110 us on the bending kernel: its `buffer_inv sc1` empties the L2 the other
// workgroups are streaming through.
struct TicketFin {
    unsigned* ticket;   // nullptr: no in-kernel finalize (the caller launches mean_final_kernel)
    float* out;
    int nout;           // batch items covered by this launch: part[b * nb + k], k < nb
    long long nb;
    double denom;
    float sign;         // -1 for -mean(cc), +1 for the bending mean; times the caller's scale
    int accumulate;
};

// this workgroup's partial: a plain store without a ticket, an agent-scope (write-through) one with
__device__ __forceinline__ void publish_partial(double* part, long long idx, double r, const TicketFin& f)
{
    if (f.ticket) __hip_atomic_store(part + idx, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else part[idx] = r;
}

// `scratch`: >= 256 B of shared memory, 8-B aligned, that no thread of the workgroup still reads or writes (the kernels pass
// their own, no longer needed, tile: a static __shared__ here would push the 80-KB NCC tile past two workgroups per CU).
// Call after thread 0 has run publish_partial.
__device__ __forceinline__ void ticket_finalize(const double* part, const TicketFin& f, char* scratch)
{
    if (f.ticket == nullptr) return;
    int& s_last = *reinterpret_cast<int*>(scratch);
    double* s_red = reinterpret_cast<double*>(scratch + 64);
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the partial has been written through ...
        const unsigned t = __hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the ticket counts it
        s_last = (t == gridDim.x * gridDim.y - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    for (int b = 0; b < f.nout; ++b) {
        double a = 0.0;
        for (long long k = threadIdx.x; k < f.nb; k += blockDim.x)
            a += __hip_atomic_load(part + (long long)b * f.nb + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double r = block_sum(a, s_red);
        if (threadIdx.x == 0) {
            const float v = (float)((double)f.sign * r / f.denom);
            f.out[b] = f.accumulate ? f.out[b] + v : v;
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(f.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------ Dice ------------------------------------ //
// y [B, nvox, L]; grid (nblk, B); threads >= T=(256/L)*L idle so that a
// thread's label is fixed while the block still reads contiguous runs.
__global__ void __launch_bounds__(RED_BLOCK)
dice_partial_kernel(const float* __restrict__ yt, const float* __restrict__ yp, double* __restrict__ part,
                    int64_t nvox, int L, int nblk, int zeropad)
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
            float a = t[e], c = p[e];
            if (zeropad) {  // losses.py:34-57: voxels whose background channel is >= 1 in either map are zeroed
                const int64_t v0 = (e / L) * L;
                if (t[v0] >= 1.f || p[v0] >= 1.f) a = c = 0.f;
            }
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

__global__ void __launch_bounds__(64)
dice_sum_kernel(const double* __restrict__ part, float* __restrict__ top_bot, int L, int nblk)
{
    const int i = blockIdx.x;  // b * L + l: one wave, ordered strided sums + wave reduction
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

__global__ void __launch_bounds__(RED_BLOCK)
dice_final_kernel(const float* __restrict__ top_bot, float* __restrict__ loss, int B, int L, int mode, int zeropad)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < B * L; i += blockDim.x) {
        const float ft = top_bot[i * 2], fb = top_bot[i * 2 + 1];
        const bool counted = !zeropad || (i / L == 0 && i % L >= 1);  // zeropad: labels 1..L-1 of batch item 0
        if (counted) acc += (double)dice_ratio(ft, fb, mode);
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = (float)(-s / (double)(zeropad ? (L - 1) : B * L));
}

// d dice / d y_pred[b,v,l] = -scale/(B L) * (2 t / bot - top / bot^2); MMR_DICE_DIVIDE_NO_NAN: 0 where bot == 0;
// MMR_DICE_MAX_EPS: bot below the clamp is the constant 1e-5 (only the 2 t / eps term remains)
__global__ void __launch_bounds__(256)
dice_bwd_kernel(const float* __restrict__ y_true, const float* __restrict__ top_bot, float* __restrict__ dpred, int B,
                int64_t nvox, int L, float scale, int accumulate, int mode)
{
    const int64_t total = (int64_t)B * nvox * L;
    const float c = -scale / (float)(B * L);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int l = (int)(i % L);
        const int b = (int)(i / (nvox * L));
        const float top = top_bot[(b * L + l) * 2], bot = top_bot[(b * L + l) * 2 + 1];
        float ga, gb;
        dice_ratio_grad(top, bot, mode, ga, gb);
        const float g = c * (gb * y_true[i] + ga);
        if (accumulate) dpred[i] += g; else dpred[i] = g;
    }
}

// ------------------------------ Grad-l2 --------------------------------- //
// A wave walks whole (x, y) rows of Z * C contiguous floats (4 rows per workgroup at a time), so the position of an element is
// one small division per element instead of the five 64-bit ones of an element-indexed loop (46 -> see DESIGN at 160^3 x 3).
__global__ void __launch_bounds__(RED_BLOCK)
grad_l2_partial_kernel(const float* __restrict__ f, double* __restrict__ part, int X, int Y, int Z, int C, int nblk)
{
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const int rowlen = Z * C;
    const int64_t sy = rowlen, sx = (int64_t)Y * rowlen;
    const float* p = f + (int64_t)b * X * sx;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float ax = 0.f, ay = 0.f, az = 0.f;
    const int nrows = X * Y;
    for (int row = blockIdx.x * 4 + wv; row < nrows; row += nblk * 4) {
        const int y = row % Y, x = row / Y;
        const float* r = p + (int64_t)row * rowlen;
        const bool hx = x + 1 < X, hy = y + 1 < Y;
        for (int t = lane; t < rowlen; t += 64) {
            const float c0 = r[t];
            if (hx) { const float d = r[t + sx] - c0; ax += d * d; }
            if (hy) { const float d = r[t + sy] - c0; ay += d * d; }
            if (t + C < rowlen) { const float d = r[t + C] - c0; az += d * d; }
        }
    }
    const double rx = block_sum((double)ax, sh);
    const double ry = block_sum((double)ay, sh);
    const double rz = block_sum((double)az, sh);
    if (threadIdx.x == 0) {
        double* o = part + ((int64_t)b * nblk + blockIdx.x) * 3;
        o[0] = rx; o[1] = ry; o[2] = rz;
    }
}

// one wave per batch item: ordered strided sums of the per-block partials + wave reduction
__global__ void __launch_bounds__(64)
grad_l2_final_kernel(const double* __restrict__ part, float* __restrict__ out, int B, int X, int Y,
                     int Z, int C, int nblk, float loss_mult)
{
    const int b = blockIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) {
        const double* o = part + ((int64_t)b * nblk + k) * 3;
        s0 += o[0]; s1 += o[1]; s2 += o[2];
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (threadIdx.x == 0) {
        const double nx = (double)(X - 1) * Y * Z * C, ny = (double)X * (Y - 1) * Z * C, nz = (double)X * Y * (Z - 1) * C;
        out[b] = (float)((s0 / nx + s1 / ny + s2 / nz) / 3.0 * (double)loss_mult);
    }
}

// ------------------------------ local NCC ------------------------------- //
// Two wave-centric passes, no LDS, no block barriers:
//  pass 1 (zy box): a wave owns a strip of rows x 56 z-outputs (64 lanes incl. 4+4 halo) of one x-plane
//    and walks the rows; the 9-wide z window is a log-step wavefront-shuffle reduction
//    (s2 = v + shfl(v,1); s4 = s2 + shfl(s2,2); s8 = s4 + shfl(s4,4); s9 = s8 + shfl(v,8)), the 9-row y
//    window a register ring.  Writes the 5 zy-box sums of (I, J, I^2, J^2, IJ).
//  pass 2 (x box + cc): a thread owns a (y,z) column and slides a 9-plane register ring along x.
// 'SAME' zero padding falls out of loading zeros outside the volume.
// Local cross-correlation of one window from its five box sums S = (sum I, sum J, sum I^2, sum J^2, sum IJ), ws = window
// size.  form: MMR_NCC_CLASSIC cc = cross^2 / (Iv Jv + eps); MMR_NCC_CLAMPED (newer voxelmorph):
// cross, Iv, Jv clamped to >= eps, cc = (cross / Iv) * (cross / Jv).  Returns cc and d cc / d (cross, Iv, Jv).
struct NccTerms { float cc, uI, uJ, A, Bc, Cc; };
// FAST: quotients through v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division sequence
template <bool FAST = false>
__device__ __forceinline__ float ncc_div(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }
template <bool FAST = false>
__device__ __forceinline__ NccTerms ncc_terms(const float* S, float ws, float eps, int form)
{
    NccTerms t;
    t.uI = FAST ? S[0] * (1.0f / ws) : S[0] / ws;
    t.uJ = FAST ? S[1] * (1.0f / ws) : S[1] / ws;
    float cross = S[4] - t.uJ * S[0] - t.uI * S[1] + t.uI * t.uJ * ws;
    float Iv = S[2] - 2.f * t.uI * S[0] + t.uI * t.uI * ws;
    float Jv = S[3] - 2.f * t.uJ * S[1] + t.uJ * t.uJ * ws;
    if (form == MMR_NCC_CLAMPED) {
        const bool kc = cross > eps, ki = Iv > eps, kj = Jv > eps;   // tf.maximum passes the gradient to the larger side
        cross = kc ? cross : eps;
        Iv = ki ? Iv : eps;
        Jv = kj ? Jv : eps;
        const float ri = ncc_div<FAST>(cross, Iv), rj = ncc_div<FAST>(cross, Jv);
        t.cc = ri * rj;
        t.A = kc ? 2.f * cross / (Iv * Jv) : 0.f;
        t.Bc = ki ? -ri * rj / Iv : 0.f;
        t.Cc = kj ? -ri * rj / Jv : 0.f;
    } else {
        const float den = Iv * Jv + eps;
        const float r = ncc_div<FAST>(cross, den);
        t.cc = cross * r;
        t.A = 2.f * r;
        t.Bc = -r * r * Jv;
        t.Cc = -r * r * Iv;
    }
    return t;
}

// Forward only: the cc of FOUR windows from their box sums S[k][0..4] = (sum I, sum J, sum I^2, sum J^2, sum IJ), two windows per
// packed instruction (v_pk_mul / v_pk_fma), with the window means eliminated algebraically:
//   cross = S4 - S0 S1 / ws,  Iv = S2 - S0^2 / ws,  Jv = S3 - S1^2 / ws
// (the same quantities as ncc_terms' uI / uJ form, A8: 6 packed + 1 reciprocal per pair instead of ~25 scalar per window).
typedef float f2_t __attribute__((ext_vector_type(2)));
template <int FORM>
__device__ __forceinline__ float ncc_cc4_sum(const float (&S)[4][5], float eps)
{
    const f2_t k = {1.0f / 729.f, 1.0f / 729.f};
    f2_t tot = {0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f2_t s0 = {S[2 * h][0], S[2 * h + 1][0]}, s1 = {S[2 * h][1], S[2 * h + 1][1]}, s2 = {S[2 * h][2], S[2 * h + 1][2]},
                   s3 = {S[2 * h][3], S[2 * h + 1][3]}, s4 = {S[2 * h][4], S[2 * h + 1][4]};
        const f2_t t0 = s0 * k, t1 = s1 * k;
        f2_t cross = __builtin_elementwise_fma(-t0, s1, s4);
        f2_t iv = __builtin_elementwise_fma(-t0, s0, s2);
        f2_t jv = __builtin_elementwise_fma(-t1, s1, s3);
        if (FORM == MMR_NCC_CLAMPED) {      // tf.maximum(., eps) on all three, cc = (cross / Iv) (cross / Jv)
            cross = f2_t{fmaxf(cross.x, eps), fmaxf(cross.y, eps)};
            iv = f2_t{fmaxf(iv.x, eps), fmaxf(iv.y, eps)};
            jv = f2_t{fmaxf(jv.x, eps), fmaxf(jv.y, eps)};
            const f2_t den = iv * jv;
            const f2_t rc = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            tot = __builtin_elementwise_fma(cross * cross, rc, tot);
        } else {
            const f2_t e2 = {eps, eps};
            const f2_t den = __builtin_elementwise_fma(iv, jv, e2);
            const f2_t rc = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            tot = __builtin_elementwise_fma(cross * cross, rc, tot);
        }
    }
    return tot.x + tot.y;
}

// Backward: the five per-window coefficient fields of FOUR windows from their box sums, means eliminated as above.  With
//   A = d cc / d cross,  Bc = d cc / d Iv,  Cc = d cc / d Jv   (ncc_terms)   and   d cross / d I_p = J_p - uJ,  d Iv / d I_p = 2 (I_p - uI):
//   d cc_c / d I_p = J_p A + I_p (2 Bc) - (A uJ + 2 Bc uI),     d cc_c / d J_p = I_p A + J_p (2 Cc) - (A uI + 2 Cc uJ)
// for every window c that contains p, i.e. FIVE fields to box-filter (A, 2 Bc, 2 Cc and the two mean terms), not seven.
typedef float f4_t __attribute__((ext_vector_type(4)));
template <int FORM>
__device__ __forceinline__ void ncc_coef4(const float (&S)[4][5], float eps, f4_t (&C)[5])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float uI = S[k][0] * (1.0f / 729.f), uJ = S[k][1] * (1.0f / 729.f);
        float cross = __builtin_fmaf(-uI, S[k][1], S[k][4]);
        float iv = __builtin_fmaf(-uI, S[k][0], S[k][2]);
        float jv = __builtin_fmaf(-uJ, S[k][1], S[k][3]);
        float A, B2, C2;
        if (FORM == MMR_NCC_CLAMPED) {     // tf.maximum passes the gradient to the larger side (ncc_terms)
            const bool kc = cross > eps, ki = iv > eps, kj = jv > eps;
            cross = kc ? cross : eps;
            iv = ki ? iv : eps;
            jv = kj ? jv : eps;
            const float qi = __builtin_amdgcn_rcpf(iv), qj = __builtin_amdgcn_rcpf(jv);
            const float ri = cross * qi, rj = cross * qj, m = -2.f * ri * rj;
            A = kc ? 2.f * ri * qj : 0.f;
            B2 = ki ? m * qi : 0.f;
            C2 = kj ? m * qj : 0.f;
        } else {
            const float r = cross * __builtin_amdgcn_rcpf(__builtin_fmaf(iv, jv, eps));
            const float m = -2.f * r * r;
            A = 2.f * r;
            B2 = m * jv;
            C2 = m * iv;
        }
        C[0][k] = A;
        C[1][k] = B2;
        C[2][k] = C2;
        C[3][k] = __builtin_fmaf(A, uJ, B2 * uI);
        C[4][k] = __builtin_fmaf(A, uI, C2 * uJ);
    }
}

constexpr int NCC_ZOUT = 56;   // outputs per wave along z (lanes 4..59)
constexpr int NCC_ROWS = 32;   // output rows per wave strip

__device__ __forceinline__ float box9_lanes(float v)
{
    const float s2 = v + __shfl_down(v, 1, 64);
    const float s4 = s2 + __shfl_down(s2, 2, 64);
    const float s8 = s4 + __shfl_down(s4, 4, 64);
    return s8 + __shfl_down(v, 8, 64);  // lane l holds sum of lanes l..l+8
}

__global__ void __launch_bounds__(256)
ncc_zybox_kernel(const float* __restrict__ I, const float* __restrict__ J, float* __restrict__ zy, int B, int X, int Y,
                 int Z, int nzs, int nys)
{
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)B * X * nys * nzs;
    if (wid >= nw) return;
    int64_t t = wid;
    const int zs = (int)(t % nzs); t /= nzs;
    const int ys = (int)(t % nys); t /= nys;
    const int x = (int)(t % X);
    const int b = (int)(t / X);
    const int z = zs * NCC_ZOUT + lane - 4;          // input z of this lane; box result at lane l covers z..z+8
    const int y0 = ys * NCC_ROWS;
    const int y1 = (y0 + NCC_ROWS < Y) ? y0 + NCC_ROWS : Y;
    const bool zin = z >= 0 && z < Z;
    const size_t nvox = (size_t)X * Y * Z;
    const float* Ip = I + (size_t)b * nvox + (size_t)x * Y * Z;
    const float* Jp = J + (size_t)b * nvox + (size_t)x * Y * Z;
    float ring[9][5];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int q = 0; q < 5; ++q) ring[k][q] = 0.f;
    // output for centre z_c = z + 4 lives in lane l (window lanes l..l+8): valid for lanes 0..55
    const int zc = z + 4;
    const bool zout = lane < NCC_ZOUT && zc < Z;
    for (int yb = y0 - 4; yb < y1 + 4; yb += 9) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = yb + k;
            if (yy < y1 + 4) {
                float a = 0.f, c = 0.f;
                if (zin && yy >= 0 && yy < Y) {
                    a = Ip[(size_t)yy * Z + z];
                    c = Jp[(size_t)yy * Z + z];
                }
                ring[k][0] = box9_lanes(a);
                ring[k][1] = box9_lanes(c);
                ring[k][2] = box9_lanes(a * a);
                ring[k][3] = box9_lanes(c * c);
                ring[k][4] = box9_lanes(a * c);
                const int yo = yy - 4;
                if (yo >= y0 && yo < y1 && zout) {
                    float* o = zy + ((((size_t)b * 5) * X + x) * Y + yo) * Z + zc;
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        float s = 0.f;
#pragma unroll
                        for (int j = 0; j < 9; ++j) s += ring[j][q];
                        o[(size_t)q * nvox] = s;
                    }
                }
            }
        }
    }
}

// zy: [B][5][X][Y][Z]; thread = (y,z) column of one x segment
__global__ void __launch_bounds__(256)
ncc_xbox_kernel(const float* __restrict__ zy, double* __restrict__ part, int X, int Y, int Z, int xseg, int nseg, float eps,
                int form)
{
    __shared__ double sh[4];
    const int b = blockIdx.z / nseg, seg = blockIdx.z % nseg;
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int ncol = Y * Z;
    const int x0 = seg * xseg;
    const int x1 = (x0 + xseg < X) ? x0 + xseg : X;
    const size_t nvox = (size_t)X * Y * Z;
    const float* base = zy + (size_t)b * 5 * nvox + col;
    const float ws = 729.f;
    float ring[9][5];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int q = 0; q < 5; ++q) ring[k][q] = 0.f;
    float acc = 0.f;
    if (col < ncol) {
        for (int xb = x0 - 4; xb < x1 + 4; xb += 9) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int xs = xb + k;
                if (xs < x1 + 4) {
                    const bool xin = xs >= 0 && xs < X;
#pragma unroll
                    for (int q = 0; q < 5; ++q) ring[k][q] = xin ? base[(size_t)q * nvox + (size_t)xs * ncol] : 0.f;
                    const int xo = xs - 4;
                    if (xo >= x0 && xo < x1) {
                        float S[5];
#pragma unroll
                        for (int q = 0; q < 5; ++q) {
                            float s = 0.f;
#pragma unroll
                            for (int j = 0; j < 9; ++j) s += ring[j][q];
                            S[q] = s;
                        }
                        acc += ncc_terms(S, ws, eps, form).cc;
                    }
                }
            }
        }
    }
    const double r = block_sum((double)acc, sh);
    if (threadIdx.x == 0) part[((size_t)b * nseg + seg) * gridDim.x + blockIdx.x] = r;
}

// ------------------------------ local NCC, single pass ------------------------------ //
// Forward only (the backward below keeps the two-pass form because it needs the window sums as fields).
// One 512-thread workgroup owns a (32 y) x (56 z) output tile and marches along an x segment; nothing but the two
// input volumes is read and nothing but one partial sum per workgroup is written (the two-pass form moved 0.8 GB at
// 256^3 for 134 MB of input).  The three box filters use three different mechanisms, so no window-sum volume exists:
//   x: every lane keeps the RAW (I, J) of the last 9 planes of its 5 haloed rows in registers (90 VGPRs); the five
//      window sums (I, J, I^2, J^2, IJ) slide (+ new plane - leaving plane) and are re-summed exactly from the ring
//      every 9th plane, so a sum never carries more than 8 add/subtract pairs of rounding (no drift along x);
//   y: the 40 haloed rows x 5 x-sums are exchanged through LDS (double buffered, one barrier per plane) and each wave
//      sums the 9-row windows of its 4 output rows;
//   z: lanes are z; the 9-wide window of those 4 x 5 values is 8 dependent `v_add_f32_dpp ... wave_shr:1` (whole-wave
//      shift by one lane, zero shifted in at lane 0): VALU only, 7 clk per wave-instruction measured
//      (tools/ubench/dpp_rate.hip) against 40 for the ds_bpermute + add the two-pass kernels use.
// 'SAME' zero padding: out-of-volume samples are loaded from a clamped (in-bounds) address and multiplied by a 0/1
// mask -- a `cond ? load : 0` makes hipcc branch around every load and wait for each one in turn.  Tile geometry:
// lanes 0..63 hold input z = 56 zt - 4 + lane; after the right-shifting box, lane l >= 8 holds the window centred
// at z - 4 = 56 zt + (l - 8) (56 outputs).
constexpr int NF_WAVES = 8;                       // waves per workgroup
constexpr int NF_RPW = 5;                         // haloed rows per wave
constexpr int NF_ROWS = NF_WAVES * NF_RPW;        // 40 haloed rows
constexpr int NF_YOUT = NF_ROWS - 8;              // 32 output rows per tile
constexpr int NF_OPW = NF_YOUT / NF_WAVES;        // 4 output rows per wave
constexpr int NF_ZOUT = 56;
constexpr int NF_LDS_BYTES = 2 * NF_ROWS * 5 * 64 * (int)sizeof(float);   // 102,400 B

__device__ __forceinline__ float wave_shr1(float v)   // lane l <- lane l-1, lane 0 <- 0
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float box9_shr(float v)    // lane l: v[l-8] + ... + v[l]
{
    float s = v;
#pragma unroll
    for (int i = 0; i < 8; ++i) s = v + wave_shr1(s);
    return s;
}

__global__ void __launch_bounds__(NF_WAVES * 64)
ncc_fused_kernel(const float* __restrict__ I, const float* __restrict__ J, double* __restrict__ part, int X, int Y, int Z,
                 int xseg, int nxs, int nyt, int nzt, float eps, int form, const TicketFin fin)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* zb = reinterpret_cast<float*>(smem);            // [2][NF_ROWS][5][64]
    __shared__ double sh[NF_WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int t = blockIdx.x;
    const int zt = t % nzt; t /= nzt;
    const int yt = t % nyt; t /= nyt;
    const int xs = t % nxs;
    const int b = t / nxs;
    const int x0 = xs * xseg;
    const int x1 = (x0 + xseg < X) ? x0 + xseg : X;
    const int z = zt * NF_ZOUT - 4 + lane;                  // input z of this lane
    const bool zin = z >= 0 && z < Z;
    const size_t nvox = (size_t)X * Y * Z;
    const float* Ib = I + (size_t)b * nvox;
    const float* Jb = J + (size_t)b * nvox;
    const int yrow0 = yt * NF_YOUT - 4 + w * NF_RPW;        // first haloed row of this wave
    int roff[NF_RPW];
    float rmask[NF_RPW];
#pragma unroll
    for (int r = 0; r < NF_RPW; ++r) {
        const int y = yrow0 + r;
        const bool in = zin && y >= 0 && y < Y;
        rmask[r] = in ? 1.f : 0.f;
        roff[r] = in ? y * Z + z : 0;
    }
    float ra[9][NF_RPW], rc[9][NF_RPW];                     // raw ring of the last 9 planes, plane s in slot s % 9
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < NF_RPW; ++r) ra[k][r] = rc[k][r] = 0.f;
    // outputs of this wave: tile rows NF_OPW*w .. +3 (haloed rows NF_OPW*w .. NF_OPW*w + 11)
    const int yo0 = yt * NF_YOUT + w * NF_OPW;
    const int zc = z - 4;
    const bool zout = lane >= 8 && zc < Z;                  // zc >= 0 follows from lane >= 8 (zt >= 0)
    const float ws = 729.f;
    float acc = 0.f;
    float pa[NF_RPW], pc[NF_RPW];                           // prefetched plane
    const int xi0 = x0 - 4, nstep = (x1 - x0) + 8;
    {
        const float xm = xi0 >= 0 ? 1.f : 0.f;              // xi0 < X always
        const size_t po = (size_t)(xi0 >= 0 ? xi0 : 0) * Y * Z;
#pragma unroll
        for (int r = 0; r < NF_RPW; ++r) {
            pa[r] = Ib[po + roff[r]] * (xm * rmask[r]);
            pc[r] = Jb[po + roff[r]] * (xm * rmask[r]);
        }
    }
    float W[NF_RPW][5];                                     // x-window sums of this lane's rows
#pragma unroll
    for (int r = 0; r < NF_RPW; ++r)
#pragma unroll
        for (int q = 0; q < 5; ++q) W[r][q] = 0.f;
    int slot = 0;                                           // ring slot of plane s = s % 9: the slot the leaving plane holds
    // consecutive all-zero planes per row (see ncc_fused4_kernel): between two exact re-sums a window that has moved from bright
    // tissue into a zero background keeps a rounding residual; with nine zero planes in the ring the sums ARE zero
    int zrun[NF_RPW];
#pragma unroll
    for (int r = 0; r < NF_RPW; ++r) zrun[r] = 0;
    for (int s = 0; s < nstep; ++s) {
#pragma unroll
        for (int r = 0; r < NF_RPW; ++r)
            zrun[r] = (((__float_as_uint(pa[r]) | __float_as_uint(pc[r])) << 1) == 0u) ? zrun[r] + 1 : 0;
        // slide the window sums (the leaving plane sits in the slot the new one takes) and take the prefetched plane.
        // The ring is addressed through a uniform switch over the slot: every case names its registers statically, so the
        // ring never moves (rotating it cost 80 v_mov per plane and wave).
        const bool refresh = slot == 8;                     // s = 8, 17, 26, ...: exact re-sum instead (uniform)
#define MMR_NCC_SLOT(K)                                                                       \
    case K: {                                                                                 \
        if (s > 8 && K != 8) {                                                                \
            _Pragma("unroll") for (int r = 0; r < NF_RPW; ++r) {                              \
                const float a = pa[r], c = pc[r], ao = ra[K][r], co = rc[K][r];               \
                W[r][0] += a - ao;                                                            \
                W[r][1] += c - co;                                                            \
                W[r][2] += a * a - ao * ao;                                                   \
                W[r][3] += c * c - co * co;                                                   \
                W[r][4] += a * c - ao * co;                                                   \
            }                                                                                 \
        }                                                                                     \
        _Pragma("unroll") for (int r = 0; r < NF_RPW; ++r) { ra[K][r] = pa[r]; rc[K][r] = pc[r]; } \
    } break;
        switch (slot) {
            MMR_NCC_SLOT(0) MMR_NCC_SLOT(1) MMR_NCC_SLOT(2) MMR_NCC_SLOT(3) MMR_NCC_SLOT(4)
            MMR_NCC_SLOT(5) MMR_NCC_SLOT(6) MMR_NCC_SLOT(7) MMR_NCC_SLOT(8)
        }
#undef MMR_NCC_SLOT
        slot = slot == 8 ? 0 : slot + 1;
        {
            const int xn = xi0 + s + 1;
            const bool xin = xn >= 0 && xn < X;             // the plane after the last step is loaded but never used
            const float xm = xin ? 1.f : 0.f;
            const size_t po = (size_t)(xin ? xn : 0) * Y * Z;
#pragma unroll
            for (int r = 0; r < NF_RPW; ++r) {
                pa[r] = Ib[po + roff[r]] * (xm * rmask[r]);
                pc[r] = Jb[po + roff[r]] * (xm * rmask[r]);
            }
        }
        if (s < 8) continue;                                // window not yet full (uniform branch)
        if (refresh) {
#pragma unroll
            for (int r = 0; r < NF_RPW; ++r) {
                float sI = 0.f, sJ = 0.f, sII = 0.f, sJJ = 0.f, sIJ = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float a = ra[k][r], c = rc[k][r];
                    sI += a; sJ += c; sII += a * a; sJJ += c * c; sIJ += a * c;
                }
                W[r][0] = sI; W[r][1] = sJ; W[r][2] = sII; W[r][3] = sJJ; W[r][4] = sIJ;
            }
        }
#pragma unroll
        for (int r = 0; r < NF_RPW; ++r)
#pragma unroll
            for (int q = 0; q < 5; ++q) W[r][q] = zrun[r] >= 9 ? 0.f : W[r][q];
        float* buf = zb + (size_t)(s & 1) * (NF_ROWS * 5 * 64);
#pragma unroll
        for (int r = 0; r < NF_RPW; ++r) {
            float* o = buf + ((w * NF_RPW + r) * 5) * 64 + lane;
#pragma unroll
            for (int q = 0; q < 5; ++q) o[q * 64] = W[r][q];
        }
        __syncthreads();
        // y windows of this wave's 4 output rows (rows j .. j+8 of the 12 it reads; common part rows 3..8), then the z box
        float S[NF_OPW][5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            float v[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = buf[((w * NF_OPW + j) * 5 + q) * 64 + lane];
            const float A = ((v[3] + v[4]) + (v[5] + v[6])) + (v[7] + v[8]);
            S[0][q] = box9_shr(A + ((v[0] + v[1]) + v[2]));
            S[1][q] = box9_shr(A + ((v[1] + v[2]) + v[9]));
            S[2][q] = box9_shr(A + ((v[2] + v[9]) + v[10]));
            S[3][q] = box9_shr(A + ((v[9] + v[10]) + v[11]));
        }
#pragma unroll
        for (int j = 0; j < NF_OPW; ++j)
            if (zout && yo0 + j < Y) acc += ncc_terms<true>(S[j], ws, eps, form).cc;
    }
    double v = wave_sum((double)acc);
    if (lane == 0) sh[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < NF_WAVES; ++i) r += sh[i];
        publish_partial(part, blockIdx.x, r, fin);
    }
    ticket_finalize(part, fin, smem);      // (its first __syncthreads separates this from the last tile reads)
}

__global__ void mean_final_kernel(const double* __restrict__ part, float* __restrict__ out, int B, int64_t nb,
                                  double denom, float sign, int accumulate = 0)
{
    __shared__ double sh[4];
    const int b = blockIdx.x;
    double a = 0.0;
    for (int64_t k = threadIdx.x; k < nb; k += blockDim.x) a += part[(int64_t)b * nb + k];
    const double s = block_sum(a, sh);
    if (threadIdx.x == 0) {
        const float v = (float)((double)sign * s / denom);
        out[b] = accumulate ? out[b] + v : v;
    }
}

// ------------------------------ bending --------------------------------- //
// Bending energy, forward.  A lane owns one voxel column position z (its three channels are one 12-byte load per
// row), a wave owns BF_R output rows (BF_R + 2 haloed) of a 64-wide z strip and marches along an x segment with the
// three planes of the stencil in registers: every input row is loaded ONCE per plane as a contiguous 768-byte wave
// load; the z +- 1 neighbours are whole-wave DPP shifts (wave_shr:1 / wave_shl:1), not loads.  The first version
// issued 27 scalar loads per voxel (1.65 TB/s at 256^3); this one issues (BF_R + 2) / BF_R.
constexpr int BF_ZOUT = 62;      // output voxels per wave along z (lanes 1..62)

__device__ __forceinline__ float wave_shl1(float v)   // lane l <- lane l+1, lane 63 <- 0
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// ------------------------------ local NCC, single pass, four z per lane ------------------------------ //
// Three mechanisms (x: sliding sums over the 9-plane window; y: LDS exchange; z: DPP).  A lane owns FOUR consecutive z (one
// 16-B load per row and plane, all plain arithmetic as packed v_pk_*), so a row of up to 256 voxels is one wave and the centred
// 9-window of a lane's four outputs needs its neighbours' values only from lanes l - 1 and l + 1:
//   out0 = P[l-1] + P + v0[l+1]      out1 = (v1+v2+v3)[l-1] + P + (v0+v1)[l+1]
//   out2 = (v2+v3)[l-1] + P + (v0+v1+v2)[l+1]      out3 = v3[l-1] + P + P[l+1]        (P = v0+v1+v2+v3)
// = 5 plain + 8 DPP additions per field for four outputs (2 DPP per output instead of 8); whole rows also mean no z halo.  A wave
// holds 2 haloed rows, the tile is 16 rows -> 8 outputs and every wave sums the 9-row y window of ONE output row from LDS -- as
// four published row PAIRS + one single row, 25 ds_read_b128 instead of 45 (80 KB tile, single buffer, two barriers per plane).
// Requires Z % 4 == 0 and Z <= 256 (else the one-z-per-lane kernel runs).
//
// Round 4: NO register ring.  Until round 3 the raw 9-plane x window of the wave's two rows lived in registers (144 of 255
// VGPRs, plane loop unrolled by nine, exact re-sum every ninth plane): two waves per SIMD, one workgroup per CU, and the
// counters said it waited (waves parked 44 % of their cycles, VALU active 24 %; profiles/r04a_ncc_pmc_sq.json).  Now the plane
// that LEAVES the window is simply loaded again, eight steps after it joined (from the Infinity Cache: both volumes are 134 MB):
// no ring, no parked sums, 128 VGPRs and exactly 80 KB of LDS -> two workgroups = four waves per SIMD on every CU; 512 workgroups
// of 16-plane x segments instead of 256 of 32 (x warm-up 1.5x instead of 1.25x).  83 -> 73 us at 256^3 (same-box A/B, same value to
// the last printed digit).  The sums slide for at most xseg + 8 <= 72 steps without a re-sum: the drift of a 9-term fp32 sum
// over that many add / subtract pairs stays two orders below the rounding of the 729-term window sum it feeds.
constexpr int N4_WAVES = 8, N4_RPW = 2, N4_ROWS = N4_WAVES * N4_RPW, N4_YOUT = N4_ROWS - 8;
constexpr int N4_TILE_BYTES = N4_ROWS * 5 * 64 * 16;                  // 81,920: the published x-sums
// y + x[lane - 1] / y + x[lane + 1] (0 beyond the wave's ends) as ONE v_add_f32_dpp.  Written out because hipcc kept the
// shifts of this kernel as v_mov_b32_dpp + a separate add (40 extra VALU per plane step); the s_nop covers the two wait
// states a DPP read needs after a VALU write of its source (the hazard recogniser does not look inside inline asm).
__device__ __forceinline__ float add_shr1(float x, float y)
{
    float d;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(x), "v"(y));
    return d;
}
__device__ __forceinline__ float add_shl1(float x, float y)
{
    float d;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(x), "v"(y));
    return d;
}

constexpr int N4_LDS_BYTES = N4_TILE_BYTES;

// the centred 9-window of a lane's four z outputs from the lane's four y-summed values v and its two neighbours' (see above):
// eight DPP-fused additions in ONE asm block -- every DPP-read source (P, s123, s23, v.w, then v.x, p01, p012, P) is computed
// before the block, so one s_nop 1 covers the two wait states a DPP read needs after a VALU write (each add_shr1 / add_shl1 on
// its own carries that nop: 40 per plane step)
__device__ __forceinline__ void zwin4(const f4_t v, float& o0, float& o1, float& o2, float& o3)
{
    const float p01 = v.x + v.y, p012 = p01 + v.z, P = p012 + v.w, s23 = v.z + v.w, s123 = v.y + s23;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %6, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %7, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %8, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %9, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %10, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %4, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
        : "v"(P), "v"(s123), "v"(s23), "v"(v.w), "v"(v.x), "v"(p01), "v"(p012));
}

// COEF (the backward's first pass): the same march, but instead of reducing cc every wave writes the five coefficient fields of
// ncc_coef4 for its output row to coef [B][5][X][Y][Z]; nothing is reduced and `part` / `fin` are unused.
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
template <int FORM, bool COEF = false>
__global__ void __launch_bounds__(N4_WAVES * 64, 4)
ncc_fused4_kernel(const float* __restrict__ I, const float* __restrict__ J, double* __restrict__ part, int X, int Y, int Z,
                   int xseg, int nxs, int nyt, float eps, const TicketFin fin, float* __restrict__ coef = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f4_t* buf = reinterpret_cast<f4_t*>(smem);             // [N4_ROWS][5][64]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // w in an SGPR: uniform branches on it
    int t = blockIdx.x;
    // workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one L2): every XCD takes a contiguous run of logical
    // workgroups, so that y-neighbouring tiles, which share 8 of their 16 haloed rows, read them through one L2
    {
        const int nwg = gridDim.x, xcd = t & 7, qd = nwg >> 3, rm = nwg & 7;
        t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (t >> 3);
    }
    const int lblk = t;
    const int yt = t % nyt; t /= nyt;
    const int xs = t % nxs;
    const int b = t / nxs;
    const int x0 = xs * xseg;
    const int x1 = (x0 + xseg < X) ? x0 + xseg : X;
    const bool zin = 4 * lane < Z;
    const size_t nvox = (size_t)X * Y * Z;
    const float* Ib = I + (size_t)b * nvox;
    const float* Jb = J + (size_t)b * nvox;
    unsigned rofs[N4_RPW];
#pragma unroll
    for (int r = 0; r < N4_RPW; ++r) {
        const int y = yt * N4_YOUT - 4 + w * N4_RPW + r;
        const bool in = zin && y >= 0 && y < Y;
        rofs[r] = in ? (unsigned)(y * Z + 4 * lane) * 4u : 0xF0000000u;
    }
    const unsigned vol_bytes = (unsigned)(nvox * 4);
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ib), 0, (int)vol_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsJ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Jb), 0, (int)vol_bytes, 0x00020000);
    // plane xq (wave-uniform), row r; `use` false or a plane outside the volume -> offset beyond num_records -> hardware zeros
    auto ldp = [&](const __amdgpu_buffer_rsrc_t& rs, int xq, int r, bool use) -> f4_t {
        const bool xin = use && xq >= 0 && xq < X;
        const unsigned soff = xin ? (unsigned)xq * (unsigned)(Y * Z) * 4u : 0u;
        const unsigned voff = xin ? rofs[r] : 0xF0000000u;
        return __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    };
    const f4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    f4_t W[N4_RPW][5];
#pragma unroll
    for (int r = 0; r < N4_RPW; ++r)
#pragma unroll
        for (int q = 0; q < 5; ++q) W[r][q] = zero4;
    // Sliding sums cannot return to EXACT zero by themselves: once a voxel column's window has left bright tissue for a zero
    // background, ((W + a) - a) keeps a rounding residual of ~1e-7 of what passed through, and for un-normalised intensities
    // (0..255, 0..4095) that residual, squared into the variances, dwarfs eps: windows that lie wholly in the background would
    // score O(1) garbage where the reference's conv-based sums are exactly 0 and cc = 0.  zrun[r] counts, per lane AND per z
    // component (a 9^3 window can contain one z column of a lane's quad without its neighbours), the consecutive planes in which
    // I and J were both zero (I^2 + J^2 == 0); the step at which the ninth such plane joins, that column's window IS all zero
    // and its five sums are set to it.  From then on 0 + 0 - 0 stays exact.  Planes outside the volume arrive as hardware zeros.
    typedef int i4_t __attribute__((ext_vector_type(4)));
    i4_t zrun[N4_RPW];
#pragma unroll
    for (int r = 0; r < N4_RPW; ++r) zrun[r] = i4_t{0, 0, 0, 0};
    auto join = [&](int r, const f4_t a, const f4_t c) {
        W[r][0] += a; W[r][1] += c; W[r][2] += a * a; W[r][3] += c * c; W[r][4] += a * c;
        const f4_t e = __builtin_elementwise_fma(c, c, a * a);
        i4_t& z = zrun[r];
        z.x = e.x == 0.f ? z.x + 1 : 0; z.y = e.y == 0.f ? z.y + 1 : 0; z.z = e.z == 0.f ? z.z + 1 : 0; z.w = e.w == 0.f ? z.w + 1 : 0;
        const bool hit = (z.x == 9) | (z.y == 9) | (z.z == 9) | (z.w == 9);
        if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {            // rare: some column's window has just become all zero
            // divergent constant stores (exec-masked v_mov), not selects: a select reads the updated sums, hipcc then keeps those in
            // fresh registers and pays ten 64-bit moves per row on the COMMON path to bring them back
            if (z.x == 9) { W[r][0].x = 0.f; W[r][1].x = 0.f; W[r][2].x = 0.f; W[r][3].x = 0.f; W[r][4].x = 0.f; }
            if (z.y == 9) { W[r][0].y = 0.f; W[r][1].y = 0.f; W[r][2].y = 0.f; W[r][3].y = 0.f; W[r][4].y = 0.f; }
            if (z.z == 9) { W[r][0].z = 0.f; W[r][1].z = 0.f; W[r][2].z = 0.f; W[r][3].z = 0.f; W[r][4].z = 0.f; }
            if (z.w == 9) { W[r][0].w = 0.f; W[r][1].w = 0.f; W[r][2].w = 0.f; W[r][3].w = 0.f; W[r][4].w = 0.f; }
        }
    };
    const bool oval = zin && (yt * N4_YOUT + w) < Y;
    float acc = 0.f;
    const int xi0 = x0 - 4, nstep = (x1 - x0) + 8;
    __amdgpu_buffer_rsrc_t rsC[5];
    unsigned cofs = 0xF0000000u;
    if (COEF) {
#pragma unroll
        for (int q = 0; q < 5; ++q)
            rsC[q] = __builtin_amdgcn_make_buffer_rsrc(coef + ((size_t)b * 5 + q) * nvox, 0, (int)vol_bytes, 0x00020000);
        if (oval) cofs = (unsigned)((yt * N4_YOUT + w) * Z + 4 * lane) * 4u;      // else beyond num_records: the store is dropped
    }
    f4_t na[N4_RPW], nc[N4_RPW], oa[N4_RPW], oc[N4_RPW];   // the plane that joins the window at this step / the one that leaves after it
    // warm-up: the eight planes in front of the first complete window join in two batches of four whose loads are all in flight
    // together (one plane per round trip, as the steady state does it, left a fresh workgroup waiting eight round trips)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f4_t pa[4][N4_RPW], pc[4][N4_RPW];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < N4_RPW; ++r) {
                pa[k][r] = ldp(rsI, xi0 + 4 * h + k, r, true);
                pc[k][r] = ldp(rsJ, xi0 + 4 * h + k, r, true);
            }
        if (h == 1) {
#pragma unroll
            for (int r = 0; r < N4_RPW; ++r) {     // the first steady step's planes ride behind the second batch
                na[r] = ldp(rsI, xi0 + 8, r, true);
                nc[r] = ldp(rsJ, xi0 + 8, r, true);
                oa[r] = ldp(rsI, xi0, r, true);
                oc[r] = ldp(rsJ, xi0, r, true);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < N4_RPW; ++r) join(r, pa[k][r], pc[k][r]);
    }
    for (int s = 8; s < nstep; ++s) {
#pragma unroll
        for (int r = 0; r < N4_RPW; ++r) join(r, na[r], nc[r]);
#pragma unroll
        for (int q = 0; q < 5; ++q) buf[(w * 5 + q) * 64 + lane] = W[0][q] + W[1][q];
        if (w < 4) {      // (a select per element costs 20 v_cndmask per step)
#pragma unroll
            for (int q = 0; q < 5; ++q) buf[((8 + w) * 5 + q) * 64 + lane] = W[1][q];
        } else {
#pragma unroll
            for (int q = 0; q < 5; ++q) buf[((8 + w) * 5 + q) * 64 + lane] = W[0][q];
        }
        // plane s - 8 leaves the window before the next step; then the next step's two planes are requested (a step ahead)
#pragma unroll
        for (int r = 0; r < N4_RPW; ++r) {
            // packed, in place, through the VOP3P neg modifiers (hipcc turns `W -= a` into four v_sub_f32 and negates the fma
            // operands with v_xor; negated copies made with v_pk_mul cost registers the kernel does not have)
            const f4_t a = oa[r], c = oc[r];
            auto sub4 = [](f4_t& Wq, const f4_t v) {
                f2_t lo = __builtin_shufflevector(Wq, Wq, 0, 1), hi = __builtin_shufflevector(Wq, Wq, 2, 3);
                const f2_t vl = __builtin_shufflevector(v, v, 0, 1), vh = __builtin_shufflevector(v, v, 2, 3);
                asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(lo) : "v"(vl));
                asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(hi) : "v"(vh));
                Wq = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
            };
            auto fnma4 = [](f4_t& Wq, const f4_t u, const f4_t v) {      // Wq -= u * v
                f2_t lo = __builtin_shufflevector(Wq, Wq, 0, 1), hi = __builtin_shufflevector(Wq, Wq, 2, 3);
                const f2_t ul = __builtin_shufflevector(u, u, 0, 1), uh = __builtin_shufflevector(u, u, 2, 3);
                const f2_t vl = __builtin_shufflevector(v, v, 0, 1), vh = __builtin_shufflevector(v, v, 2, 3);
                asm("v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(lo) : "v"(ul), "v"(vl));
                asm("v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(hi) : "v"(uh), "v"(vh));
                Wq = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
            };
            sub4(W[r][0], a); sub4(W[r][1], c); fnma4(W[r][2], a, a); fnma4(W[r][3], c, c); fnma4(W[r][4], a, c);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < N4_RPW; ++r) {
            na[r] = ldp(rsI, xi0 + s + 1, r, true);
            nc[r] = ldp(rsJ, xi0 + s + 1, r, true);
            oa[r] = ldp(rsI, xi0 + s - 7, r, true);
            oc[r] = ldp(rsJ, xi0 + s - 7, r, true);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        const int p0 = (w + 1) >> 1;
        const int s1 = 8 + ((w & 1) ? (w >> 1) : (w >> 1) + 4);
        float S[4][5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            f4_t v = buf[(s1 * 5 + q) * 64 + lane];
#pragma unroll
            for (int k = 0; k < 4; ++k) v += buf[((p0 + k) * 5 + q) * 64 + lane];
            zwin4(v, S[0][q], S[1][q], S[2][q], S[3][q]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (COEF) {
            f4_t C[5];
            ncc_coef4<FORM>(S, eps, C);
            const unsigned soff = (unsigned)(x0 + s - 8) * (unsigned)(Y * Z) * 4u;       // the window's centre plane
#pragma unroll
            for (int q = 0; q < 5; ++q)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, C[q]), rsC[q], cofs, soff, 0);
        } else if (oval) {
            acc += ncc_cc4_sum<FORM>(S, eps);
        }
        __syncthreads();                                    // the tile is rewritten by the next plane
    }
    if (COEF) return;
    double v = wave_sum((double)acc);
    double* sh = reinterpret_cast<double*>(smem);
    if (lane == 0) sh[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < N4_WAVES; ++i) r += sh[i];
        publish_partial(part, lblk, r, fin);
    }
    ticket_finalize(part, fin, smem + 1024);
}

// The backward's second pass: the 9^3 box filter (zero padded) of the coefficient fields, with the SAME march as the forward (x:
// sliding sums with the leaving plane loaded again; y: pair sums through LDS; z: DPP), combined at the voxel with
// I_p, J_p into the gradients:   dI_p = s (J_p [A] + I_p [2 Bc] - [A uJ + 2 Bc uI]),   dJ_p = s (I_p [A] + J_p [2 Cc] - [A uI + 2 Cc uJ]),
// s = -gout[b] / N, [.] = box sum.  NF = 3: ONE gradient (its three fields; `only` = 0: dI, 1: dJ; a doubled grid, only = -1, gives
// both and was measured slower than NF = 5), every load of the next step in flight a step ahead as in the forward.  NF = 5: both gradients
// from one filter of all five fields; 2 rows x 5 fields x (joining + leaving plane) would be 80 registers of loads in flight, so
// the leaving plane is requested into the joining plane's registers once those have been added (two round trips per step, the
// second one under the y / z sums).  Sliding sums of coefficient fields return to exact zero the same way as the forward's: a
// per-lane, per-z counter of consecutive planes in which every field was zero.
struct NccGradArgs { const float* C; const float* I; const float* J; const float* gout; float* dI; float* dJ; };

template <int NF>
__global__ void __launch_bounds__(N4_WAVES * 64, 4)
ncc_boxgrad4_kernel(const NccGradArgs a, int X, int Y, int Z, int xseg, int nxs, int nyt, int nblk, int only)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f4_t* buf = reinterpret_cast<f4_t*>(smem);             // [N4_ROWS][NF][64]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int t = blockIdx.x;
    {   // XCD-contiguous runs of logical workgroups (see ncc_fused4_kernel)
        const int nwg = gridDim.x, xcd = t & 7, qd = nwg >> 3, rm = nwg & 7;
        t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (t >> 3);
    }
    const int which = NF == 5 ? 0 : (only >= 0 ? only : t / nblk);      // NF == 3: 0 = dI, 1 = dJ
    t %= nblk;
    const int yt = t % nyt; t /= nyt;
    const int xs = t % nxs;
    const int b = t / nxs;
    const int x0 = xs * xseg;
    const int x1 = (x0 + xseg < X) ? x0 + xseg : X;
    const bool zin = 4 * lane < Z;
    const size_t nvox = (size_t)X * Y * Z;
    unsigned rofs[N4_RPW];
#pragma unroll
    for (int r = 0; r < N4_RPW; ++r) {
        const int y = yt * N4_YOUT - 4 + w * N4_RPW + r;
        const bool in = zin && y >= 0 && y < Y;
        rofs[r] = in ? (unsigned)(y * Z + 4 * lane) * 4u : 0xF0000000u;
    }
    const int yo = yt * N4_YOUT + w;
    const unsigned oofs = (zin && yo < Y) ? (unsigned)(yo * Z + 4 * lane) * 4u : 0xF0000000u;   // this wave's output row
    const unsigned vol_bytes = (unsigned)(nvox * 4);
    const unsigned plane_bytes = (unsigned)(Y * Z) * 4u;
    auto rsrc = [&](const float* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)vol_bytes, 0x00020000); };
    __amdgpu_buffer_rsrc_t rsC[NF];
    const float* Cb = a.C + (size_t)b * 5 * nvox;
#pragma unroll
    for (int q = 0; q < NF; ++q) {
        const int f = NF == 5 ? q : (q == 0 ? 0 : (q == 1 ? 1 + which : 3 + which));
        rsC[q] = rsrc(Cb + (size_t)f * nvox);
    }
    // P multiplies [A], Q the second field: (J, I) for dI, (I, J) for dJ; with NF == 5: P = J, Q = I
    const __amdgpu_buffer_rsrc_t rsP = rsrc((which == 0 ? a.J : a.I) + (size_t)b * nvox);
    const __amdgpu_buffer_rsrc_t rsQ = rsrc((which == 0 ? a.I : a.J) + (size_t)b * nvox);
    const __amdgpu_buffer_rsrc_t rsO0 = rsrc((which == 0 ? a.dI : a.dJ) + (size_t)b * nvox);
    const __amdgpu_buffer_rsrc_t rsO1 = rsrc((NF == 5 ? a.dJ : a.dI) + (size_t)b * nvox);       // used with NF == 5 only
    const float sc = -(a.gout ? a.gout[b] : 1.f) / (float)nvox;          // loss_b = -mean(cc)
    auto ldp = [&](const __amdgpu_buffer_rsrc_t& rs, int xq, unsigned row) -> f4_t {
        const bool xin = xq >= 0 && xq < X;
        const unsigned soff = xin ? (unsigned)xq * plane_bytes : 0u;
        const unsigned voff = xin ? row : 0xF0000000u;
        return __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    };
    const f4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    f4_t W[N4_RPW][NF];
#pragma unroll
    for (int r = 0; r < N4_RPW; ++r)
#pragma unroll
        for (int q = 0; q < NF; ++q) W[r][q] = zero4;
    typedef int i4_t __attribute__((ext_vector_type(4)));
    i4_t zrun[N4_RPW];
#pragma unroll
    for (int r = 0; r < N4_RPW; ++r) zrun[r] = i4_t{0, 0, 0, 0};
    auto join = [&](int r, const f4_t (&v)[NF]) {
        i4_t any = __builtin_bit_cast(i4_t, v[0]);
#pragma unroll
        for (int q = 0; q < NF; ++q) W[r][q] += v[q];
#pragma unroll
        for (int q = 1; q < NF; ++q) any |= __builtin_bit_cast(i4_t, v[q]);
        any &= 0x7fffffff;
        i4_t& z = zrun[r];
        z.x = any.x == 0 ? z.x + 1 : 0; z.y = any.y == 0 ? z.y + 1 : 0; z.z = any.z == 0 ? z.z + 1 : 0; z.w = any.w == 0 ? z.w + 1 : 0;
        const bool hit = (z.x == 9) | (z.y == 9) | (z.z == 9) | (z.w == 9);
        if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {            // rare: some column's window has just become all zero
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                if (z.x == 9) W[r][q].x = 0.f;
                if (z.y == 9) W[r][q].y = 0.f;
                if (z.z == 9) W[r][q].z = 0.f;
                if (z.w == 9) W[r][q].w = 0.f;
            }
        }
    };
    auto sub4 = [](f4_t& Wq, const f4_t v) {      // packed, in place (see ncc_fused4_kernel)
        f2_t lo = __builtin_shufflevector(Wq, Wq, 0, 1), hi = __builtin_shufflevector(Wq, Wq, 2, 3);
        const f2_t vl = __builtin_shufflevector(v, v, 0, 1), vh = __builtin_shufflevector(v, v, 2, 3);
        asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(lo) : "v"(vl));
        asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(hi) : "v"(vh));
        Wq = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
    };
    const int xi0 = x0 - 4, nstep = (x1 - x0) + 8;
    constexpr bool TWOPHASE = NF == 5;
    constexpr int WB = NF == 5 ? 1 : 2;                    // planes per warm-up batch (their loads are in flight together)
    f4_t nw[N4_RPW][NF], ol[TWOPHASE ? 1 : N4_RPW][NF];    // the joining plane / the leaving one (TWOPHASE: `nw` serves both)
    f4_t ip, iq;                                           // the images at this wave's output row, a step ahead
#pragma unroll 1
    for (int h = 0; h < 8 / WB; ++h) {
        f4_t pv[WB][N4_RPW][NF];
#pragma unroll
        for (int k = 0; k < WB; ++k)
#pragma unroll
            for (int r = 0; r < N4_RPW; ++r)
#pragma unroll
                for (int q = 0; q < NF; ++q) pv[k][r][q] = ldp(rsC[q], xi0 + WB * h + k, rofs[r]);
#pragma unroll
        for (int k = 0; k < WB; ++k)
#pragma unroll
            for (int r = 0; r < N4_RPW; ++r) join(r, pv[k][r]);
    }
#pragma unroll
    for (int r = 0; r < N4_RPW; ++r)
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            nw[r][q] = ldp(rsC[q], xi0 + 8, rofs[r]);
            if (!TWOPHASE) ol[r][q] = ldp(rsC[q], xi0, rofs[r]);
        }
    ip = ldp(rsP, x0, oofs);
    iq = ldp(rsQ, x0, oofs);
    for (int s = 8; s < nstep; ++s) {
#pragma unroll
        for (int r = 0; r < N4_RPW; ++r) join(r, nw[r]);
        if (TWOPHASE) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < N4_RPW; ++r)
#pragma unroll
                for (int q = 0; q < NF; ++q) nw[r][q] = ldp(rsC[q], xi0 + s - 8, rofs[r]);      // the plane that leaves after this step
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < NF; ++q) buf[(w * NF + q) * 64 + lane] = W[0][q] + W[1][q];
        if (w < 4) {
#pragma unroll
            for (int q = 0; q < NF; ++q) buf[((8 + w) * NF + q) * 64 + lane] = W[1][q];
        } else {
#pragma unroll
            for (int q = 0; q < NF; ++q) buf[((8 + w) * NF + q) * 64 + lane] = W[0][q];
        }
        const f4_t pcur = ip, qcur = iq;
        if (!TWOPHASE) {
#pragma unroll
            for (int r = 0; r < N4_RPW; ++r)
#pragma unroll
                for (int q = 0; q < NF; ++q) sub4(W[r][q], ol[r][q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < N4_RPW; ++r)
#pragma unroll
                for (int q = 0; q < NF; ++q) {
                    nw[r][q] = ldp(rsC[q], xi0 + s + 1, rofs[r]);
                    ol[r][q] = ldp(rsC[q], xi0 + s - 7, rofs[r]);
                }
            ip = ldp(rsP, x0 + s - 7, oofs);
            iq = ldp(rsQ, x0 + s - 7, oofs);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        const int p0 = (w + 1) >> 1;
        const int s1 = 8 + ((w & 1) ? (w >> 1) : (w >> 1) + 4);
        float S[4][NF];
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            f4_t v = buf[(s1 * NF + q) * 64 + lane];
#pragma unroll
            for (int k = 0; k < 4; ++k) v += buf[((p0 + k) * NF + q) * 64 + lane];
            zwin4(v, S[0][q], S[1][q], S[2][q], S[3][q]);
            __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned soff = (unsigned)(x0 + s - 8) * plane_bytes;
        f4_t g0, g1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (NF == 5) {
                g0[k] = sc * (pcur[k] * S[k][0] + qcur[k] * S[k][1] - S[k][3]);
                g1[k] = sc * (qcur[k] * S[k][0] + pcur[k] * S[k][2] - S[k][4]);
            } else {
                g0[k] = sc * (pcur[k] * S[k][0] + qcur[k] * S[k][1] - S[k][2]);
            }
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, g0), rsO0, oofs, soff, 0);
        if (NF == 5) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, g1), rsO1, oofs, soff, 0);
        __syncthreads();                                    // the tile is rewritten by the next plane
        if (TWOPHASE) {
#pragma unroll
            for (int r = 0; r < N4_RPW; ++r)
#pragma unroll
                for (int q = 0; q < NF; ++q) sub4(W[r][q], nw[r][q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < N4_RPW; ++r)
#pragma unroll
                for (int q = 0; q < NF; ++q) nw[r][q] = ldp(rsC[q], xi0 + s + 1, rofs[r]);
            ip = ldp(rsP, x0 + s - 7, oofs);
            iq = ldp(rsQ, x0 + s - 7, oofs);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// x[lane + 1] + x[lane - 1] and x[lane + 1] - x[lane - 1] (0 beyond the wave's ends): one DPP move + one DPP-fused VOP2
// instead of two moves and an add (hipcc keeps the second shift as its own v_mov_b32_dpp); the s_nop covers the two wait
// states a DPP read needs after a VALU write of its source
__device__ __forceinline__ float shl_plus_shr(float x)
{
    const float t = wave_shl1(x);
    float d;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(x), "v"(t));
    return d;
}
__device__ __forceinline__ float shl_minus_shr(float x)
{
    const float t = wave_shl1(x);
    float d;   // v_subrev: dst = src1 - dpp(src0)
    asm("s_nop 1\n\tv_subrev_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(x), "v"(t));
    return d;
}

struct F3 { float v[3]; };

template <int BF_R>              // output rows per wave (BF_R + 2 rows loaded per plane)
__global__ void __launch_bounds__(RED_BLOCK, 2)
bending_fused_kernel(const float* __restrict__ u, double* __restrict__ part, int X, int Y, int Z, int xseg, int nxs, int nyg,
                     int nzs, int64_t nwaves, const TicketFin fin)
{
    __shared__ double sh[4];
    __shared__ __attribute__((aligned(8))) char fin_scratch[256];
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (an XCD-contiguous workgroup order measured no better)
    float acc = 0.f;
    if (wid < nwaves) {
        // y row groups fastest: the four waves of a workgroup are y-neighbours of one (z strip, x segment), so the two halo
        // rows a wave shares with the next one are served by the CU's L1 instead of crossing the fabric twice
        int64_t t = wid;
        const int yg = (int)(t % nyg); t /= nyg;
        const int zs = (int)(t % nzs); t /= nzs;
        const int xs = (int)(t % nxs);
        const int b = (int)(t / nxs);
        const int z = zs * BF_ZOUT + lane;                  // outputs at lanes 1..62 -> z = zs*62 + 1 .. zs*62 + 62
        const bool zin = z < Z;
        const int yh0 = yg * BF_R;                          // first haloed row; output rows yh0 + 1 .. yh0 + BF_R
        const int xa = 1 + xs * xseg;
        const int xb = (xa + xseg < X - 1) ? xa + xseg : X - 1;
        const size_t sx = (size_t)Y * Z * 3;
        const float* base = u + (size_t)b * X * sx;           // wave-uniform; per-lane parts are 32-bit offsets
        const unsigned zoff = (unsigned)(zin ? z : 0) * 3u;   // clamped: masked lanes load in bounds
        unsigned roff[BF_R + 2];
        float rmask[BF_R + 2];
#pragma unroll
        for (int r = 0; r < BF_R + 2; ++r) {
            const bool in = zin && (yh0 + r) < Y;
            rmask[r] = in ? 1.f : 0.f;
            roff[r] = (in ? (unsigned)(yh0 + r) * (unsigned)Z * 3u : 0u) + zoff;
        }
        F3 p0[BF_R + 2], p1[BF_R + 2], p2[BF_R + 2], pn[BF_R + 2];
        auto load_plane = [&](F3* p, int x) {
            const float* q = base + (size_t)x * sx;
#pragma unroll
            for (int r = 0; r < BF_R + 2; ++r) {
                // one 12-byte load from an always-in-bounds address, zeroed by a multiplicative mask: a `cond ? load : 0`
                // makes hipcc branch around every load and wait for each in turn (8 dependent round trips per plane)
                const F3 t = *reinterpret_cast<const F3*>(q + roff[r]);
#pragma unroll
                for (int k = 0; k < 3; ++k) p[r].v[k] = t.v[k] * rmask[r];
            }
        };
        load_plane(p0, xa - 1);
        load_plane(p1, xa);
        load_plane(p2, xa + 1);
        const bool zout = lane >= 1 && lane <= BF_ZOUT && z <= Z - 2;
        // one stencil step on planes (q0, q1, q2) = x - 1, x, x + 1 while plane x + 2 lands in qn.  The x loop is unrolled by
        // four with the plane names rotated, so the four plane sets never move (rotating them cost 54 v_mov per step).
        auto step = [&](const F3* q0, const F3* q1, const F3* q2, F3* qn, int x) {
            load_plane(qn, (x + 2 < X) ? x + 2 : X - 1);     // one plane ahead of the stencil: its latency hides under this step
#pragma unroll
            for (int j = 0; j < BF_R; ++j) {
                float e = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float c0 = q1[j + 1].v[k];
                    const float dxx = q2[j + 1].v[k] - 2.f * c0 + q0[j + 1].v[k];
                    const float dyy = q1[j + 2].v[k] - 2.f * c0 + q1[j].v[k];
                    const float dzz = shl_plus_shr(c0) - 2.f * c0;
                    const float dxy = (q2[j + 2].v[k] - q2[j].v[k] - q0[j + 2].v[k] + q0[j].v[k]) * 0.25f;
                    // mixed z differences: difference across x (or y) first, then ONE pair of lane shifts of it
                    const float dx = q2[j + 1].v[k] - q0[j + 1].v[k];
                    const float dy = q1[j + 2].v[k] - q1[j].v[k];
                    const float dxz = shl_minus_shr(dx) * 0.25f;
                    const float dyz = shl_minus_shr(dy) * 0.25f;
                    e += dxx * dxx + dyy * dyy + dzz * dzz + 2.f * (dxy * dxy + dxz * dxz + dyz * dyz);
                }
                if (zout && (yh0 + 1 + j) <= Y - 2) acc += e;
            }
        };
        for (int x = xa; x < xb; x += 4) {
            step(p0, p1, p2, pn, x);
            if (x + 1 < xb) step(p1, p2, pn, p0, x + 1);
            if (x + 2 < xb) step(p2, pn, p0, p1, x + 2);
            if (x + 3 < xb) step(pn, p0, p1, p2, x + 3);
        }
    }
    const double r = block_sum((double)acc, sh);
    if (threadIdx.x == 0) publish_partial(part, blockIdx.x, r, fin);
    ticket_finalize(part, fin, fin_scratch);
}

// ------------------------------ NCC backward ---------------------------- //
// With S = box sums of (I, J, I^2, J^2, IJ) at window centre c and cc = cross^2 / (Iv Jv + eps):
//   A = d cc/d cross = 2 cross / (Iv Jv + eps),  Bc = d cc/d Iv = -cross^2 Jv / (Iv Jv + eps)^2,  Cc likewise with Iv;
//   d cross / d I_p = J_p - uJ(c),  d Iv / d I_p = 2 (I_p - uI(c))   for every window c that contains p, hence
//   dL/dI_p = -g/N [ J_p box(A) - box(A uJ) + 2 I_p box(Bc) - 2 box(Bc uI) ],  dL/dJ_p symmetric,
// i.e. seven more 9^3 box filters (zero padded like the forward) of per-window coefficient fields.
// pass 2': the x box of the forward sums, writing the seven coefficient fields instead of reducing cc
__global__ void __launch_bounds__(256)
ncc_xcoef_kernel(const float* __restrict__ zy, float* __restrict__ F, int X, int Y, int Z, int xseg, int nseg, float eps,
                 int form)
{
    const int b = blockIdx.z / nseg, seg = blockIdx.z % nseg;
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int ncol = Y * Z;
    if (col >= ncol) return;
    const int x0 = seg * xseg;
    const int x1 = (x0 + xseg < X) ? x0 + xseg : X;
    const size_t nvox = (size_t)X * Y * Z;
    const float* base = zy + (size_t)b * 5 * nvox + col;
    float* out = F + (size_t)b * 7 * nvox + col;
    const float ws = 729.f;
    float ring[9][5];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int q = 0; q < 5; ++q) ring[k][q] = 0.f;
    for (int xb = x0 - 4; xb < x1 + 4; xb += 9) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int xs = xb + k;
            if (xs < x1 + 4) {
                const bool xin = xs >= 0 && xs < X;
#pragma unroll
                for (int q = 0; q < 5; ++q) ring[k][q] = xin ? base[(size_t)q * nvox + (size_t)xs * ncol] : 0.f;
                const int xo = xs - 4;
                if (xo >= x0 && xo < x1) {
                    float S[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        float t = 0.f;
#pragma unroll
                        for (int j = 0; j < 9; ++j) t += ring[j][q];
                        S[q] = t;
                    }
                    const NccTerms t = ncc_terms(S, ws, eps, form);
                    const float A = t.A, Bc = t.Bc, Cc = t.Cc, uI = t.uI, uJ = t.uJ;
                    float* o = out + (size_t)xo * ncol;
                    o[0] = A; o[nvox] = A * uJ; o[2 * nvox] = A * uI;
                    o[3 * nvox] = Bc; o[4 * nvox] = Bc * uI; o[5 * nvox] = Cc; o[6 * nvox] = Cc * uJ;
                }
            }
        }
    }
}

// z + y box of NF stored fields (same wave layout as ncc_zybox_kernel): F [B][NF][X][Y][Z] -> zy [B][NF][X][Y][Z]
template <int NF>
__global__ void __launch_bounds__(256)
box_zy_kernel(const float* __restrict__ F, float* __restrict__ zy, int B, int X, int Y, int Z, int nzs, int nys)
{
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)B * X * nys * nzs;
    if (wid >= nw) return;
    int64_t t = wid;
    const int zs = (int)(t % nzs); t /= nzs;
    const int ys = (int)(t % nys); t /= nys;
    const int x = (int)(t % X);
    const int b = (int)(t / X);
    const int z = zs * NCC_ZOUT + lane - 4;
    const int y0 = ys * NCC_ROWS;
    const int y1 = (y0 + NCC_ROWS < Y) ? y0 + NCC_ROWS : Y;
    const bool zin = z >= 0 && z < Z;
    const size_t nvox = (size_t)X * Y * Z;
    const float* Fp = F + (size_t)b * NF * nvox + (size_t)x * Y * Z;
    float ring[9][NF];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int q = 0; q < NF; ++q) ring[k][q] = 0.f;
    const int zc = z + 4;
    const bool zout = lane < NCC_ZOUT && zc < Z;
    for (int yb = y0 - 4; yb < y1 + 4; yb += 9) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = yb + k;
            if (yy < y1 + 4) {
                const bool in = zin && yy >= 0 && yy < Y;
#pragma unroll
                for (int q = 0; q < NF; ++q) ring[k][q] = box9_lanes(in ? Fp[(size_t)q * nvox + (size_t)yy * Z + z] : 0.f);
                const int yo = yy - 4;
                if (yo >= y0 && yo < y1 && zout) {
                    float* o = zy + ((((size_t)b * NF) * X + x) * Y + yo) * Z + zc;
#pragma unroll
                    for (int q = 0; q < NF; ++q) {
                        float sum = 0.f;
#pragma unroll
                        for (int j = 0; j < 9; ++j) sum += ring[j][q];
                        o[(size_t)q * nvox] = sum;
                    }
                }
            }
        }
    }
}

// x box of the seven zy-filtered coefficient fields, combined with I_p, J_p into the two gradients
__global__ void __launch_bounds__(256)
ncc_xgrad_kernel(const float* __restrict__ zy, const float* __restrict__ I, const float* __restrict__ J,
                 const float* __restrict__ gout, float* __restrict__ dI, float* __restrict__ dJ, int X, int Y, int Z,
                 int xseg, int nseg)
{
    const int b = blockIdx.z / nseg, seg = blockIdx.z % nseg;
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int ncol = Y * Z;
    if (col >= ncol) return;
    const int x0 = seg * xseg;
    const int x1 = (x0 + xseg < X) ? x0 + xseg : X;
    const size_t nvox = (size_t)X * Y * Z;
    const float* base = zy + (size_t)b * 7 * nvox + col;
    const float scale = -(gout ? gout[b] : 1.f) / (float)nvox;  // loss_b = -mean(cc)
    float ring[9][7];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int q = 0; q < 7; ++q) ring[k][q] = 0.f;
    for (int xb = x0 - 4; xb < x1 + 4; xb += 9) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int xs = xb + k;
            if (xs < x1 + 4) {
                const bool xin = xs >= 0 && xs < X;
#pragma unroll
                for (int q = 0; q < 7; ++q) ring[k][q] = xin ? base[(size_t)q * nvox + (size_t)xs * ncol] : 0.f;
                const int xo = xs - 4;
                if (xo >= x0 && xo < x1) {
                    float S[7];
#pragma unroll
                    for (int q = 0; q < 7; ++q) {
                        float t = 0.f;
#pragma unroll
                        for (int j = 0; j < 9; ++j) t += ring[j][q];
                        S[q] = t;
                    }
                    const size_t o = (size_t)b * nvox + (size_t)xo * ncol + col;
                    const float ip = I[o], jp = J[o];
                    if (dI) dI[o] = scale * (jp * S[0] - S[1] + 2.f * ip * S[3] - 2.f * S[4]);
                    if (dJ) dJ[o] = scale * (ip * S[0] - S[2] + 2.f * jp * S[5] - 2.f * S[6]);
                }
            }
        }
    }
}

// ------------------------------ bending backward ------------------------ //
// E = 1/M sum_{c interior,k} dxx^2 + dyy^2 + dzz^2 + 2 (dxy^2 + dxz^2 + dyz^2).  Until round 4 a gather kernel recomputed every
// second difference that touches an output from global memory:
// ~75 loads and ~1 000 lane-instructions per output element, 1.38 ms at 256^3 = 0.04 of the HBM roofline for a 403-MB
// read-and-write.  Tiled form (round 5): a workgroup owns a 4 x 8 x 32 tile and puts the field with a halo of 2 into LDS once
// ([k][x][y][z] planes, 41 KB: three workgroups per CU); every thread owns the x column of four outputs at its (y, z).
//   dE/du = 1/M [ 2 sum_pure D^T Q + 4 sum_mixed D^T Q ],  Q = D u where the centre is an interior voxel of the volume, else 0
// (the energy sums interior centres only); D^T has D's stencil (pure: symmetric; mixed, with its 1/4: symmetric under
// (a, b) -> (-a, -b)).  Two paths, chosen per tile (uniform):
//   * tiles at least two voxels from every face: every Q they touch is unmasked and D^T D collapses to ONE 25-point stencil on u
//       g = 39 u(0) - 8 sum_{+-1 on an axis} u + sum_{+-2 on an axis} u + 1/4 sum_{(+-2, +-2) in a coordinate plane} u
//     (2 (1, -4, 6, -4, 1) per axis; 4 / 16 (corners - 2 edge midpoints + 4 centre) per axis pair);
//   * the others: the 21 masked second differences that touch an output, each recomputed from the LDS tile.
constexpr int BB_TX = 4, BB_TY = 8, BB_TZ = 32;
constexpr int BB_UX = BB_TX + 4, BB_UY = BB_TY + 4, BB_UZ = BB_TZ + 4;      // 8 x 12 x 36 voxels with the halo
constexpr int BB_PZ = 40;                                                    // z pitch of a row in LDS (see the lane map in the kernel)
constexpr int BB_UN = BB_UX * BB_UY * BB_PZ;                                 // 3 840
constexpr int BB_LDS = 3 * BB_UN * (int)sizeof(float);                       // 46 080 B: three workgroups per CU

__global__ void __launch_bounds__(256, 3)
bending_bwd_tiled_kernel(const float* __restrict__ u, const float* __restrict__ gout, float* __restrict__ du, int X, int Y, int Z,
                         int ntx, int nty, int ntz, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* su = reinterpret_cast<float*>(smem);          // [3][BB_UN]
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tzi = t % ntz; t /= ntz;
    const int tyi = t % nty; t /= nty;
    const int txi = t % ntx;
    const int b = t / ntx;
    const int x0 = txi * BB_TX, y0 = tyi * BB_TY, z0 = tzi * BB_TZ;
    const size_t nvox = (size_t)X * Y * Z;
    const float* ub = u + (size_t)b * nvox * 3;
    struct F3 { float v[3]; };
    // field + halo 2: one 12-B load per voxel from a clamped, always valid address, zeroed outside the volume afterwards (never
    // read by a Q whose centre is interior).  A thread keeps ONE z of the 36 and walks the 96 (x, y) rows seven at a time (252 of
    // the 256 threads): the z part of the address and the bounds test is computed once, a row costs a handful of integer
    // instructions (decomposing a linear index per item cost ~80 and was as much VALU work as the stencils), and all 14 loads
    // of a thread are issued before the first is used (as a loop they ran as dependent round trips, ~14 us per tile).
    constexpr int NROW = BB_UX * BB_UY, RPP = 256 / BB_UZ, NLD = (NROW + RPP - 1) / RPP;     // 96 rows, 7 per pass, 14 passes
    const int zq = tid % BB_UZ, r0 = tid / BB_UZ;
    const int gzl = z0 + zq - 2;
    const bool zin = gzl >= 0 && gzl < Z && r0 < RPP;
    const int czl = min(max(gzl, 0), Z - 1);
    F3 ld[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int r = min(it * RPP + r0, NROW - 1);
        const int hx = r / BB_UY, hy = r - hx * BB_UY;
        const int cx = min(max(x0 + hx - 2, 0), X - 1), cy = min(max(y0 + hy - 2, 0), Y - 1);
        ld[it] = *reinterpret_cast<const F3*>(ub + (((size_t)cx * Y + cy) * Z + czl) * 3);
    }
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int r = it * RPP + r0;
        if (r0 < RPP && r < NROW) {
            const int hx = r / BB_UY, hy = r - hx * BB_UY;
            const int gx = x0 + hx - 2, gy = y0 + hy - 2;
            const bool in = zin && gx >= 0 && gx < X && gy >= 0 && gy < Y;
            const int i = r * BB_PZ + zq;
            su[i] = in ? ld[it].v[0] : 0.f;
            su[BB_UN + i] = in ? ld[it].v[1] : 0.f;
            su[2 * BB_UN + i] = in ? ld[it].v[2] : 0.f;
        }
    }
    // Lane map: a wave = all 8 y rows x 8 z (wave w takes z 8 w .. 8 w + 7), not 2 rows x 32 z.  The path is then chosen per WAVE: in
    // a tile at a z face of the volume (a quarter of all tiles at 256^3, tiles being 32 long in z) only the wave column that
    // touches the face takes the masked path, the other three the 25-point stencil.  With the row pitch 40 the 32 lanes of a
    // half-wave (4 rows x 8 z) hit banks 8 row + z: distinct.
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty = (tid >> 3) & 7, tz = wq * 8 + (tid & 7);
    constexpr int SX = BB_UY * BB_PZ, SY = BB_PZ;          // strides of su in floats (z stride 1)
    float acc[BB_TX][3];
    __syncthreads();
    const int zw0 = z0 + wq * 8;                         // first z of this wave
    const bool deep = x0 >= 2 && x0 + BB_TX <= X - 2 && y0 >= 2 && y0 + BB_TY <= Y - 2 && zw0 >= 2 && zw0 + 8 <= Z - 2;
    if (deep) {
        // x columns of the tile + halo at the 13 (y, z) offsets the stencil touches, read ONCE per channel for the thread's four
        // outputs (72 LDS reads issued together, then arithmetic; written per output hipcc interleaved 131 reads with 69 waits)
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            const float* s = su + k * BB_UN + (ty + 2) * BB_PZ + (tz + 2);      // (hx = 0, y, z); output xi sits at hx = xi + 2
            float c0[8], yp2[8], ym2[8], zp2[8], zm2[8], yp1[4], ym1[4], zp1[4], zm1[4], pp[4], pm[4], mp[4], mm[4];
#pragma unroll
            for (int hx = 0; hx < 8; ++hx) {
                c0[hx] = s[hx * SX];
                yp2[hx] = s[hx * SX + 2 * SY]; ym2[hx] = s[hx * SX - 2 * SY];
                zp2[hx] = s[hx * SX + 2]; zm2[hx] = s[hx * SX - 2];
            }
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                const int o = (xi + 2) * SX;
                yp1[xi] = s[o + SY]; ym1[xi] = s[o - SY]; zp1[xi] = s[o + 1]; zm1[xi] = s[o - 1];
                pp[xi] = s[o + 2 * SY + 2]; pm[xi] = s[o + 2 * SY - 2]; mp[xi] = s[o - 2 * SY + 2]; mm[xi] = s[o - 2 * SY - 2];
            }
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                const int h = xi + 2;
                const float ax1 = (c0[h + 1] + c0[h - 1]) + (yp1[xi] + ym1[xi]) + (zp1[xi] + zm1[xi]);
                const float ax2 = (c0[h + 2] + c0[h - 2]) + (yp2[h] + ym2[h]) + (zp2[h] + zm2[h]);
                const float cxy = (yp2[h + 2] + ym2[h + 2]) + (yp2[h - 2] + ym2[h - 2]);
                const float cxz = (zp2[h + 2] + zm2[h + 2]) + (zp2[h - 2] + zm2[h - 2]);
                const float cyz = (pp[xi] + pm[xi]) + (mp[xi] + mm[xi]);
                const float g = 39.f * c0[h] - 8.f * ax1 + ax2 + 0.25f * ((cxy + cxz) + cyz);
#pragma unroll
                for (int k2 = 0; k2 < 3; ++k2)
                    if (k2 == k) acc[xi][k2] = g;
            }
        }
    } else {
        // interior flags of the centres at offsets -1, 0, +1 from an output, per axis (bit o + 1)
        const int gy = y0 + ty, gz = z0 + tz;
        unsigned my = 0, mz = 0;
#pragma unroll
        for (int o = -1; o <= 1; ++o) {
            my |= (gy + o >= 1 && gy + o < Y - 1) ? 1u << (o + 1) : 0u;
            mz |= (gz + o >= 1 && gz + o < Z - 1) ? 1u << (o + 1) : 0u;
        }
#pragma unroll 1
        for (int xi = 0; xi < BB_TX; ++xi) {   // (not unrolled: 21 stencils x 3 channels x 4 outputs in flight spill)
            const int gx = x0 + xi;
            unsigned mx = 0;
#pragma unroll
            for (int o = -1; o <= 1; ++o) mx |= (gx + o >= 1 && gx + o < X - 1) ? 1u << (o + 1) : 0u;
            // 1.0 where the centre at offset (ox, oy, oz) is an interior voxel, else 0.0 -- a FACTOR, not a branch: behind
            // `m ? stencil : 0` hipcc put each of the 21 stencils into its own exec-masked block with its own LDS wait, 252
            // dependent round trips per tile (boundary tiles ran 8x longer than interior ones)
            auto inside = [&](int ox, int oy, int oz) {
                return ((mx >> (ox + 1)) & (my >> (oy + 1)) & (mz >> (oz + 1)) & 1u) ? 1.f : 0.f;
            };
#pragma unroll 1
            for (int k = 0; k < 3; ++k) {
                const float* s = su + k * BB_UN + ((xi + 2) * BB_UY + (ty + 2)) * BB_PZ + (tz + 2);
                // masked pure second difference along stride a at centre offset c (in floats), factor m
                auto daa = [&](int c, int a, float m) { return m * (s[c + a] - 2.f * s[c] + s[c - a]); };
                auto dab = [&](int c, int a, int bq, float m) {
                    return (0.25f * m) * (s[c + a + bq] - s[c + a - bq] - s[c - a + bq] + s[c - a - bq]);
                };
                float g = 0.f;
                g += 2.f * (daa(-SX, SX, inside(-1, 0, 0)) - 2.f * daa(0, SX, inside(0, 0, 0)) + daa(SX, SX, inside(1, 0, 0)));
                g += 2.f * (daa(-SY, SY, inside(0, -1, 0)) - 2.f * daa(0, SY, inside(0, 0, 0)) + daa(SY, SY, inside(0, 1, 0)));
                g += 2.f * (daa(-1, 1, inside(0, 0, -1)) - 2.f * daa(0, 1, inside(0, 0, 0)) + daa(1, 1, inside(0, 0, 1)));
                g += dab(-SX - SY, SX, SY, inside(-1, -1, 0)) - dab(-SX + SY, SX, SY, inside(-1, 1, 0)) -
                     dab(SX - SY, SX, SY, inside(1, -1, 0)) + dab(SX + SY, SX, SY, inside(1, 1, 0));
                g += dab(-SX - 1, SX, 1, inside(-1, 0, -1)) - dab(-SX + 1, SX, 1, inside(-1, 0, 1)) -
                     dab(SX - 1, SX, 1, inside(1, 0, -1)) + dab(SX + 1, SX, 1, inside(1, 0, 1));
                g += dab(-SY - 1, SY, 1, inside(0, -1, -1)) - dab(-SY + 1, SY, 1, inside(0, -1, 1)) -
                     dab(SY - 1, SY, 1, inside(0, 1, -1)) + dab(SY + 1, SY, 1, inside(0, 1, 1));
#pragma unroll
                for (int a2 = 0; a2 < BB_TX; ++a2)
#pragma unroll
                    for (int k2 = 0; k2 < 3; ++k2)
                        if (a2 == xi && k2 == k) acc[a2][k2] = g;     // static register names (a dynamic index would go to scratch)
            }
        }
    }
    const double M = (double)(X - 2) * (Y - 2) * (Z - 2) * 3;
    const float sc = (float)((double)(gout ? gout[b] : 1.f) / M);
    const int gy = y0 + ty, gz = z0 + tz;
    if (gy < Y && gz < Z) {
#pragma unroll
        for (int xi = 0; xi < BB_TX; ++xi) {
            const int gx = x0 + xi;
            if (gx < X) {
                float* o = du + ((size_t)b * nvox + ((size_t)gx * Y + gy) * Z + gz) * 3;
                F3 r;
                if (accumulate) {
                    const F3 old = *reinterpret_cast<const F3*>(o);
                    r.v[0] = old.v[0] + sc * acc[xi][0]; r.v[1] = old.v[1] + sc * acc[xi][1]; r.v[2] = old.v[2] + sc * acc[xi][2];
                } else {
                    r.v[0] = sc * acc[xi][0]; r.v[1] = sc * acc[xi][1]; r.v[2] = sc * acc[xi][2];
                }
                *reinterpret_cast<F3*>(o) = r;
            }
        }
    }
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
    return (int64_t)B * red_blocks(nvox * L) * L * 2 * sizeof(double) + (int64_t)B * L * 2 * sizeof(float);
}

static int dice_fwd_impl(const float* y_true, const float* y_pred, float* loss_out, float* top_bot, void* ws, int B,
                         int64_t nvox, int L, int dice_mode, int zeropad, void* stream);

extern "C" int mmr_dice_fwd_f32(const float* y_true, const float* y_pred, float* loss_out, float* top_bot, void* ws,
                                int B, int64_t nvox, int L, int dice_mode, void* stream)
{
    return dice_fwd_impl(y_true, y_pred, loss_out, top_bot, ws, B, nvox, L, dice_mode, 0, stream);
}

// losses.dice_loss_zeropad on dense (one-hot / soft) maps, as its docstring intends (losses.py:13-21,34-69; the
// reference function itself always raises): background-masked Dice of labels 1..L-1 of batch item 0
extern "C" int mmr_dice_zeropad_fwd_f32(const float* y_true, const float* y_pred, float* loss_out, float* top_bot,
                                        void* ws, int B, int64_t nvox, int L, int dice_mode, void* stream)
{
    if (L < 2) return MMR_EINVAL;
    return dice_fwd_impl(y_true, y_pred, loss_out, top_bot, ws, B, nvox, L, dice_mode, 1, stream);
}

static int dice_fwd_impl(const float* y_true, const float* y_pred, float* loss_out, float* top_bot, void* ws, int B,
                         int64_t nvox, int L, int dice_mode, int zeropad, void* stream)
{
    if (!y_true || !y_pred || !loss_out || !ws || B < 1 || nvox < 1 || L < 1 || L > RED_BLOCK) return MMR_EINVAL;
    if (dice_mode != MMR_DICE_DIVIDE_NO_NAN && dice_mode != MMR_DICE_MAX_EPS) return MMR_EINVAL;
    const int nblk = red_blocks(nvox * L);
    hipLaunchKernelGGL(dice_partial_kernel, dim3(nblk, B), dim3(RED_BLOCK), 2 * RED_BLOCK * sizeof(float),
                       as_stream(stream), y_true, y_pred, (double*)ws, nvox, L, nblk, zeropad);
    int rc = check_launch();
    if (rc) return rc;
    float* tb = top_bot ? top_bot : (float*)((char*)ws + (size_t)B * nblk * L * 2 * sizeof(double));
    hipLaunchKernelGGL(dice_sum_kernel, dim3(B * L), dim3(64), 0, as_stream(stream), (const double*)ws, tb, L, nblk);
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(dice_final_kernel, dim3(1), dim3(RED_BLOCK), 0, as_stream(stream), (const float*)tb, loss_out, B, L,
                       dice_mode, zeropad);
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
    hipLaunchKernelGGL(grad_l2_final_kernel, dim3(B), dim3(64), 0, as_stream(stream), (const double*)ws,
                       out, B, X, Y, Z, C, nblk, loss_mult);
    return check_launch();
}

namespace {
constexpr int NCC_XSEG = 64;
inline void ncc_geom(int X, int Y, int Z, int& nseg, int& ncolblk)
{
    nseg = (X + NCC_XSEG - 1) / NCC_XSEG;
    ncolblk = (int)(((int64_t)Y * Z + 255) / 256);
}
// single-pass forward: (y,z) tiles x x-segments; the segment count fills the 256 CUs once (one workgroup per CU by
// LDS) without making segments so short that their 8-plane halo dominates
inline void ncc_fused_geom(int B, int X, int Y, int Z, int& nzt, int& nyt, int& nxs, int& xseg)
{
    nzt = (Z + NF_ZOUT - 1) / NF_ZOUT;
    nyt = (Y + NF_YOUT - 1) / NF_YOUT;
    const int64_t tiles = (int64_t)B * nzt * nyt;
    int want = (int)((256 + tiles - 1) / tiles);            // segments so that tiles * segs >= ~256
    if (tiles * want > 256 && want > 1) --want;             // ... but not more than one round of workgroups
    const int max_by_len = X / 16 > 0 ? X / 16 : 1;         // keep segments >= 16 planes (halo <= 1.5x)
    nxs = want < 1 ? 1 : (want > max_by_len ? max_by_len : want);
    xseg = (X + nxs - 1) / nxs;
    nxs = (X + xseg - 1) / xseg;
}
}  // namespace

namespace {
// four-z-per-lane kernel: whole rows per wave (Z <= 256, Z % 4 == 0), 8 output rows per tile, two workgroups per CU; x segments
// of 16 .. 64 planes (shorter ones only repeat the 8-plane warm-up; longer ones would let the sliding sums drift)
inline bool ncc_fused4_ok(int64_t nvox, int Z) { return Z % 4 == 0 && Z <= 256 && nvox * 4 < 0xF0000000ll; }
inline void ncc_fused4_geom(int B, int X, int Y, int& nyt, int& nxs, int& xseg, int wgs = 512)
{
    nyt = (Y + N4_YOUT - 1) / N4_YOUT;
    const int64_t tiles = (int64_t)B * nyt;
    int want = (int)((wgs + tiles - 1) / tiles);
    if (tiles * want > wgs && want > 1) --want;
    const int max_by_len = X / 16 > 0 ? X / 16 : 1;
    nxs = want < 1 ? 1 : (want > max_by_len ? max_by_len : want);
    if ((X + nxs - 1) / nxs > 64) nxs = (X + 63) / 64;
    xseg = (X + nxs - 1) / nxs;
    nxs = (X + xseg - 1) / xseg;
}
}  // namespace

// workspace = one double per workgroup of the single-pass kernel (the larger of the two geometries)
extern "C" int64_t mmr_ncc_ws_bytes(int B, int X, int Y, int Z)
{
    if (B < 1 || X < 1 || Y < 1 || Z < 1) return MMR_EINVAL;
    int nzt, nyt, nxs, xseg;
    ncc_fused_geom(B, X, Y, Z, nzt, nyt, nxs, xseg);
    int64_t n = (int64_t)B * nzt * nyt * nxs;
    if (ncc_fused4_ok((int64_t)X * Y * Z, Z)) {
        ncc_fused4_geom(B, X, Y, nyt, nxs, xseg);
        const int64_t n4 = (int64_t)B * nyt * nxs;
        if (n4 > n) n = n4;
    }
    return n * (int64_t)sizeof(double);
}

static int ncc_fwd_impl(const float* I, const float* J, float* out, void* ws, unsigned* ticket, int B, int X, int Y, int Z,
                        int win, float eps, int ncc_form, float scale, int accumulate, void* stream)
{
    if (!I || !J || !out || !ws || B < 1 || X < 1 || Y < 1 || Z < 1) return MMR_EINVAL;
    if (ncc_form != MMR_NCC_CLASSIC && ncc_form != MMR_NCC_CLAMPED) return MMR_EINVAL;
    if (win != 9) return MMR_EUNSUPPORTED;
    if ((int64_t)Y * Z > 0x7fffffff) return MMR_EINVAL;
    int nzt, nyt, nxs, xseg;
    TicketFin fin;
    fin.ticket = ticket;
    fin.out = out;
    fin.nout = B;
    fin.denom = (double)X * Y * Z;
    fin.sign = -scale;
    fin.accumulate = accumulate;
    if (ncc_fused4_ok((int64_t)X * Y * Z, Z)) {
        ncc_fused4_geom(B, X, Y, nyt, nxs, xseg);
        const int64_t nblk4 = (int64_t)nyt * nxs;            // per batch item
        if ((int64_t)B * nblk4 > 0x7fffffff) return MMR_EINVAL;
        static bool attr4 = false;
        if (!attr4) {
            for (const void* k : {reinterpret_cast<const void*>(ncc_fused4_kernel<MMR_NCC_CLASSIC>),
                                  reinterpret_cast<const void*>(ncc_fused4_kernel<MMR_NCC_CLAMPED>)}) {
                hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, N4_LDS_BYTES);
                if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
            }
            attr4 = true;
        }
        double* part4 = (double*)ws;
        fin.nb = nblk4;
        if (ncc_form == MMR_NCC_CLAMPED)
            hipLaunchKernelGGL(ncc_fused4_kernel<MMR_NCC_CLAMPED>, dim3((unsigned)(B * nblk4)), dim3(N4_WAVES * 64), N4_LDS_BYTES,
                               as_stream(stream), I, J, part4, X, Y, Z, xseg, nxs, nyt, eps, fin);
        else
            hipLaunchKernelGGL(ncc_fused4_kernel<MMR_NCC_CLASSIC>, dim3((unsigned)(B * nblk4)), dim3(N4_WAVES * 64), N4_LDS_BYTES,
                               as_stream(stream), I, J, part4, X, Y, Z, xseg, nxs, nyt, eps, fin);
        int rc4 = check_launch();
        if (rc4 || ticket) return rc4;
        hipLaunchKernelGGL(mean_final_kernel, dim3(B), dim3(RED_BLOCK), 0, as_stream(stream), (const double*)part4, out, B,
                           nblk4, (double)X * Y * Z, -scale, accumulate);
        return check_launch();
    }
    ncc_fused_geom(B, X, Y, Z, nzt, nyt, nxs, xseg);
    const int64_t nblk = (int64_t)nzt * nyt * nxs;          // per batch item
    if ((int64_t)B * nblk > 0x7fffffff) return MMR_EINVAL;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ncc_fused_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, NF_LDS_BYTES);
        if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
        attr_set = true;
    }
    double* part = (double*)ws;
    fin.nb = nblk;
    hipLaunchKernelGGL(ncc_fused_kernel, dim3((unsigned)(B * nblk)), dim3(NF_WAVES * 64), NF_LDS_BYTES, as_stream(stream), I, J,
                       part, X, Y, Z, xseg, nxs, nyt, nzt, eps, ncc_form, fin);
    int rc = check_launch();
    if (rc || ticket) return rc;
    hipLaunchKernelGGL(mean_final_kernel, dim3(B), dim3(RED_BLOCK), 0, as_stream(stream), (const double*)part, out, B, nblk,
                       (double)X * Y * Z, -scale, accumulate);
    return check_launch();
}

extern "C" int mmr_ncc_fwd_f32(const float* I, const float* J, float* out, void* ws, int B, int X, int Y, int Z,
                               int win, float eps, int ncc_form, void* stream)
{
    return ncc_fwd_impl(I, J, out, ws, nullptr, B, X, Y, Z, win, eps, ncc_form, 1.0f, 0, stream);
}

extern "C" int mmr_ncc_fwd_ticket_f32(const float* I, const float* J, float* out, void* ws, unsigned* ticket, int B, int X, int Y,
                                      int Z, int win, float eps, int ncc_form, float scale, int accumulate, void* stream)
{
    if (!ticket) return MMR_EINVAL;
    return ncc_fwd_impl(I, J, out, ws, ticket, B, X, Y, Z, win, eps, ncc_form, scale, accumulate, stream);
}

namespace {
// waves over (z strips, y row groups, x segments, batch); x segments of 16 planes (2-plane halo) give >= 2048 waves at 256^3
constexpr int BF_ROWS = 4;    // 82 us at 256^3 against 135 us with 6 rows (190 VGPRs)
inline void bend_geom(int B, int X, int Y, int Z, int& nzs, int& nyg, int& nxs, int& xseg, int64_t& nwaves, int R = BF_ROWS)
{
    nzs = (Z - 2 + BF_ZOUT - 1) / BF_ZOUT;
    nyg = (Y - 2 + R - 1) / R;
    xseg = 16;   // measured at 256^3: 8 planes 72 us, 16: 57, 32: 73, 64: 88
    nxs = (X - 2 + xseg - 1) / xseg;
    nwaves = (int64_t)B * nzs * nyg * nxs;
}
}  // namespace

extern "C" int64_t mmr_bending_ws_bytes(int B, int X, int Y, int Z)
{
    if (B < 1 || X < 3 || Y < 3 || Z < 3) return MMR_EINVAL;
    int nzs, nyg, nxs, xseg;
    int64_t nwaves;
    bend_geom(B, X, Y, Z, nzs, nyg, nxs, xseg, nwaves, 4);     // the smallest row group = the most blocks
    return ((nwaves + 3) / 4 + B) * (int64_t)sizeof(double);
}

static int bending_fwd_impl(const float* flow, float* out, void* ws, unsigned* ticket, int B, int X, int Y, int Z, float scale,
                            int accumulate, void* stream)
{
    if (!flow || !out || !ws || B < 1 || X < 3 || Y < 3 || Z < 3) return MMR_EINVAL;
    if ((int64_t)Y * Z * 3 > 0x7fffffff) return MMR_EINVAL;
    int nzs, nyg, nxs, xseg;
    int64_t nwaves;
    bend_geom(1, X, Y, Z, nzs, nyg, nxs, xseg, nwaves, BF_ROWS);  // per batch item, so that a block never straddles items
    const int64_t nblk = (nwaves + 3) / 4;
    if (nblk * B > 0x7fffffff) return MMR_EINVAL;
    const int64_t n = (int64_t)(X - 2) * (Y - 2) * (Z - 2) * 3;
    for (int b = 0; b < B; ++b) {
        double* part = (double*)ws + (int64_t)b * nblk;
        TicketFin fin;       // one launch per batch item: each finalizes its own out[b] (the launches are ordered by the stream)
        fin.ticket = ticket;
        fin.out = out + b;
        fin.nout = 1;
        fin.nb = nblk;
        fin.denom = (double)n;
        fin.sign = scale;
        fin.accumulate = accumulate;
        hipLaunchKernelGGL(bending_fused_kernel<BF_ROWS>, dim3((unsigned)nblk), dim3(RED_BLOCK), 0, as_stream(stream),
                               flow + (size_t)b * X * Y * Z * 3, part, X, Y, Z, xseg, nxs, nyg, nzs, nwaves, fin);
        int rc = check_launch();
        if (rc) return rc;
    }
    if (ticket) return MMR_OK;
    hipLaunchKernelGGL(mean_final_kernel, dim3(B), dim3(RED_BLOCK), 0, as_stream(stream), (const double*)ws, out, B, nblk,
                       (double)n, scale, accumulate);
    return check_launch();
}

extern "C" int mmr_bending_fwd_f32(const float* flow, float* out, void* ws, int B, int X, int Y, int Z, void* stream)
{
    return bending_fwd_impl(flow, out, ws, nullptr, B, X, Y, Z, 1.0f, 0, stream);
}

extern "C" int mmr_bending_fwd_ticket_f32(const float* flow, float* out, void* ws, unsigned* ticket, int B, int X, int Y, int Z,
                                          float scale, int accumulate, void* stream)
{
    if (!ticket) return MMR_EINVAL;
    return bending_fwd_impl(flow, out, ws, ticket, B, X, Y, Z, scale, accumulate, stream);
}

// d(-mean cc)/dI and /dJ, scaled by gout[b] (null = 1); dI / dJ may be null.  Workspace: the five coefficient fields of the
// two-pass form (Z % 4 == 0, Z <= 256), else 19 volumes of fp32 for the separable four-launch form.
extern "C" int64_t mmr_ncc_bwd_ws_bytes(int B, int X, int Y, int Z)
{
    if (B < 1 || X < 1 || Y < 1 || Z < 1) return MMR_EINVAL;
    const int nvol = ncc_fused4_ok((int64_t)X * Y * Z, Z) ? 5 : 19;
    return (int64_t)B * nvol * X * Y * Z * (int64_t)sizeof(float);
}

namespace {
template <int FORM>
int ncc_bwd_fused(const float* I, const float* J, const float* gout, float* dI, float* dJ, float* coef, int B, int X, int Y, int Z,
                  float eps, hipStream_t st)
{
    int nyt, nxs, xseg;
    ncc_fused4_geom(B, X, Y, nyt, nxs, xseg);
    const int64_t nblk = (int64_t)B * nyt * nxs;
    if (2 * nblk > 0x7fffffff || (int64_t)Y * Z > 0x3fffffff) return MMR_EINVAL;
    static bool attr = false;
    if (!attr) {
        const struct { const void* k; int bytes; } ks[] = {
            {reinterpret_cast<const void*>(ncc_fused4_kernel<FORM, true>), N4_LDS_BYTES},
            {reinterpret_cast<const void*>(ncc_boxgrad4_kernel<3>), N4_ROWS * 3 * 64 * 16},
            {reinterpret_cast<const void*>(ncc_boxgrad4_kernel<5>), N4_ROWS * 5 * 64 * 16}};
        for (const auto& k : ks) {
            hipError_t e = hipFuncSetAttribute(k.k, hipFuncAttributeMaxDynamicSharedMemorySize, k.bytes);
            if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
        }
        attr = true;
    }
    TicketFin fin{};
    hipLaunchKernelGGL((ncc_fused4_kernel<FORM, true>), dim3((unsigned)nblk), dim3(N4_WAVES * 64), N4_LDS_BYTES, st, I, J,
                       (double*)nullptr, X, Y, Z, xseg, nxs, nyt, eps, fin, coef);
    const NccGradArgs a{coef, I, J, gout, dI, dJ};
    const bool both = dI && dJ;
    // both: one filter of five fields, 309 us at 256^3 against 343 us for two of three (profiles/r05_ab_ncc_bwd_two_pass.log)
    if (both)
        hipLaunchKernelGGL(ncc_boxgrad4_kernel<5>, dim3((unsigned)nblk), dim3(N4_WAVES * 64), N4_ROWS * 5 * 64 * 16, st, a, X, Y, Z,
                           xseg, nxs, nyt, (int)nblk, -1);
    else
        hipLaunchKernelGGL(ncc_boxgrad4_kernel<3>, dim3((unsigned)nblk), dim3(N4_WAVES * 64), N4_ROWS * 3 * 64 * 16, st, a, X, Y, Z,
                           xseg, nxs, nyt, (int)nblk, dI ? 0 : 1);
    return check_launch();
}
}  // namespace

extern "C" int mmr_ncc_bwd_f32(const float* I, const float* J, const float* gout, float* dI, float* dJ, void* ws, int B,
                               int X, int Y, int Z, int win, float eps, int ncc_form, void* stream)
{
    if (!I || !J || !ws || (!dI && !dJ) || B < 1 || X < 1 || Y < 1 || Z < 1) return MMR_EINVAL;
    if (ncc_form != MMR_NCC_CLASSIC && ncc_form != MMR_NCC_CLAMPED) return MMR_EINVAL;
    if (win != 9) return MMR_EUNSUPPORTED;
    if (ncc_fused4_ok((int64_t)X * Y * Z, Z))
        return ncc_form == MMR_NCC_CLAMPED
                   ? ncc_bwd_fused<MMR_NCC_CLAMPED>(I, J, gout, dI, dJ, (float*)ws, B, X, Y, Z, eps, as_stream(stream))
                   : ncc_bwd_fused<MMR_NCC_CLASSIC>(I, J, gout, dI, dJ, (float*)ws, B, X, Y, Z, eps, as_stream(stream));
    int nseg, ncolblk;
    ncc_geom(X, Y, Z, nseg, ncolblk);
    if ((int64_t)B * nseg > 65535) return MMR_EINVAL;
    const size_t nv = (size_t)B * X * Y * Z;
    float* zy5 = (float*)ws;
    float* F = zy5 + 5 * nv;
    float* zy7 = F + 7 * nv;
    const int nzs = (Z + NCC_ZOUT - 1) / NCC_ZOUT, nys = (Y + NCC_ROWS - 1) / NCC_ROWS;
    const int64_t nb1 = ((int64_t)B * X * nys * nzs + 3) / 4;
    if (nb1 > 0x7fffffff) return MMR_EINVAL;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(ncc_zybox_kernel, dim3((unsigned)nb1), dim3(256), 0, st, I, J, zy5, B, X, Y, Z, nzs, nys);
    hipLaunchKernelGGL(ncc_xcoef_kernel, dim3(ncolblk, 1, B * nseg), dim3(256), 0, st, (const float*)zy5, F, X, Y, Z,
                       NCC_XSEG, nseg, eps, ncc_form);
    hipLaunchKernelGGL(box_zy_kernel<7>, dim3((unsigned)nb1), dim3(256), 0, st, (const float*)F, zy7, B, X, Y, Z, nzs, nys);
    hipLaunchKernelGGL(ncc_xgrad_kernel, dim3(ncolblk, 1, B * nseg), dim3(256), 0, st, (const float*)zy7, I, J, gout, dI, dJ,
                       X, Y, Z, NCC_XSEG, nseg);
    return check_launch();
}

extern "C" int mmr_bending_bwd_f32(const float* flow, const float* gout, float* dflow, int B, int X, int Y, int Z,
                                   int accumulate, void* stream)
{
    if (!flow || !dflow || B < 1 || X < 3 || Y < 3 || Z < 3) return MMR_EINVAL;
    const int ntx = (X + BB_TX - 1) / BB_TX, nty = (Y + BB_TY - 1) / BB_TY, ntz = (Z + BB_TZ - 1) / BB_TZ;
    const int64_t nblk = (int64_t)B * ntx * nty * ntz;
    if (nblk > 0x7fffffff) return MMR_EINVAL;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bending_bwd_tiled_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS);
        if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL(bending_bwd_tiled_kernel, dim3((unsigned)nblk), dim3(256), BB_LDS, as_stream(stream), flow, gout, dflow, X, Y,
                       Z, ntx, nty, ntz, accumulate);
    return check_launch();
}

// gradient of mmr_dice_fwd_f32's loss w.r.t. y_pred, from the (top, bot) sums the forward returned
extern "C" int mmr_dice_bwd_f32(const float* y_true, const float* top_bot, float* dpred, int B, int64_t nvox, int L,
                                float scale, int accumulate, int dice_mode, void* stream)
{
    if (!y_true || !top_bot || !dpred || B < 1 || nvox < 1 || L < 1) return MMR_EINVAL;
    if (dice_mode != MMR_DICE_DIVIDE_NO_NAN && dice_mode != MMR_DICE_MAX_EPS) return MMR_EINVAL;
    hipLaunchKernelGGL(dice_bwd_kernel, dim3(stream_grid((int64_t)B * nvox * L, 256)), dim3(256), 0, as_stream(stream),
                       y_true, top_bot, dpred, B, nvox, L, scale, accumulate, dice_mode);
    return check_launch();
}
