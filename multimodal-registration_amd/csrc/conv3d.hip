// Conv3D(k=3,'same',stride 1) + bias + LeakyReLU for VxmDense's U-Net as an
// implicit GEMM on the gfx950 matrix cores (reference call sites:
// train_synthmorph.py:296, 3d_reg.py:305; semantics SURVEY.md Appendix A1).
//
//   M = output voxels (a 4x8x8 or 8x8x8 tile = 256 / 512 rows per workgroup)
//   N = Cout tile (BN = 32*WN*NT: 256 / 128 / 64 / 32)
//   K = 27 taps x Cin, walked as  slice (128 B of channels) -> tap -> k-step
//
// Data movement per workgroup (512 threads = 8 waves, 1 workgroup / CU):
//   * A: the haloed input tile ((TXT+2)x10x10 voxels x 128 B of channels) is staged ONCE per channel slice into
//     LDS -- bf16 / exact fp32 by LDS-DMA (global_load_lds_dwordx4, swizzle applied to the SOURCE chunk, out-of-
//     volume rows from a zero page), fp32x3 through registers because the hi/lo split happens on the way -- as
//     unpadded 128-B rows whose 16-B chunks are XOR-swizzled on the halo coordinates (swz / swz16: conflict-free
//     ds_read_b128 for every tap); all 27 taps read it with a constant per-tap address offset -> every input
//     byte is fetched from L2/HBM 2.3x (halo) instead of 27x.  The loader folds UpSampling3D(2) (nearest) and the
//     skip concatenation, so neither tensor is ever materialised.
//   * B: weights are pre-packed (mmr_conv3d_k3_pack) into the exact LDS image
//     [tile][slice][tap][16B-chunk][cout][16 B]; each tap's BN x 128 B block is streamed by LDS-DMA into a double
//     buffer while the previous tap computes (BN = 256 bf16: issued by waves 0-3 from two points INSIDE their
//     MFMA stream, see MIDDMA).
//   * MFMA: bf16 / fp32x3 -> v_mfma_f32_16x16x32_bf16 with the product formed transposed (weights as the A
//     operand) so that a lane owns 16 consecutive couts of one voxel; fp32 -> v_mfma_f32_32x32x2_f32 (exact fp32
//     FMA chain); BN = 32 -> v_mfma_f32_32x32x16_bf16.  All element types share the byte-level layout (a 16-B
//     chunk = 8 bf16, 4 fp32, or 8 hi / 8 lo halves), so one kernel template serves them.
// Variants that were measured and rejected (XCD-aware tile order, split / top-of-tap weight-DMA placements, two
// taps per barrier, s_setprio switching, fragment reuse across dz, ...) are recorded in DESIGN.md 2.2 with their
// numbers and live in the git history only.
#include "common.hpp"

#include <stdlib.h>

namespace mmr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TX = 4, TY = 8, TZ = 8;                 // output tile
constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2;  // halo tile
constexpr int HROWS = HX * HY * HZ;                   // 600
constexpr int ROWB = 128;                             // 128 B of channels, XOR-swizzled chunks (no pad)
constexpr int A_BYTES = HROWS * ROWB;                 // 86400
constexpr int CONV_THREADS = 512;

// 16 zero bytes that out-of-volume halo rows are DMA'd from (CV_DMA_A)
__device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};

struct ConvParams {
    const char* in0;
    const char* in1;
    const char* wp;
    const float* bias;
    char* out;
    int B, X, Y, Z;
    int C0, C1, up0;
    int Cout;  // real output channels (row stride of out)
    int leaky;
    float alpha;
    int out_f32;
    int ntx, nty, ntz;
    // masked-gradient epilogue (training dgrad): out *= (ymask < 0 ? alpha : 1) and the per-block column sums of
    // the masked output go to part[blockIdx.x][Cout] -- i.e. the LeakyReLU backward + bias gradient of the layer
    // that PRODUCED this conv's input, fused so that the gradient never makes a separate elementwise pass
    const float* ymask;
    double* part;
    // split-K for launches with too few workgroups to fill the chip (the 1/8- and 1/16-resolution U-Net levels, small
    // volumes): blockIdx.z = kz handles the (slice, tap) steps [kz*gsplit, (kz+1)*gsplit) and stores its raw fp32
    // partial tile to kpart[kz]; a fixed-order finalize kernel adds them, then bias / LeakyReLU / store (reproducible).
    float* kpart;
    int gsplit;
    // split store (16x16x32 kernels, training dgrad of a concat layer): output channels [0, csplit) go to `out` with
    // row stride csplit, channels [csplit, Cout) to `out1` with row stride Cout - csplit; ymask / part then refer to
    // the second range only (the skip tensor's LeakyReLU backward + bias sums)
    char* out1;
    int csplit;
    // CV_CINIT: the accumulators start from this fp32 tensor [B,X,Y,Z,Cout] instead of zero -- the second half of a
    // folded-upsample layer (see CV_UPFOLD): out = act(cinit + conv(in) + bias)
    const float* cinit = nullptr;
    // Tail launch (launch_conv): the workgroups of this launch are the tiles tile0 .. tile0 + gridDim.x - 1; with kcompact the
    // split-K partials go to a compact buffer kpart[kz][blockIdx.x][tile voxel][Cout] (conv_ktail_finalize_kernel reads it)
    int tile0 = 0;
    int kcompact = 0;
    int xcd_pair = 0;   // CV_UPFOLD: parity classes per XCD group, 0 = blockIdx.y order (see the kernel)
    // CV_POOLF: gradient of MaxPooling3D(2) of ymask's tensor, [B, X/2, Y/2, Z/2, Cout] (see CV_POOLF)
    const float* dpool = nullptr;
};

template <int N> struct IntTag { static constexpr int value = N; };

// Variant bits of conv3d_k3_kernel's VAR parameter (the per-tile defaults are chosen in dispatch_conv)
constexpr int CV_M16 = 1 << 5;      // 16x16x32 MFMA with the transposed product (every dtype but exact fp32, N tiles >= 64)
constexpr int CV_PIPE = 1 << 7;     // fragment schedule of a tap written out with sched_group_barrier
constexpr int CV_DMA_A = 1 << 8;    // bf16 / exact fp32: the haloed A tile goes global -> LDS by DMA
constexpr int CV_STAMP = 1 << 10;   // cycle stamps (only with -DMMR_DIAG; never the measured build)
constexpr int CV_BATCHA = 1 << 12;  // fp32x3 / x1: all staging loads of a slice issued branch-free, masked when stored
constexpr int CV_PRIO_Y = 1 << 15;  // static s_setprio 1 for waves 4-7
// Masked data gradient (training) of a tensor y that ALSO feeds a MaxPooling3D(2): the gradient that arrives through the pooling
// (dpool, at half resolution) is routed to the first maximum of every 2x2x2 window of y and added in this epilogue, before the
// LeakyReLU mask and the bias column sums -- instead of a separate pass that reads y and reads + rewrites the whole gradient
// (maxpool_bwd_v4_kernel: 3.3 GB for the first skip tensor of a 160^3 step, 0.6 ms).  The epilogue already loads y at every
// voxel for the mask; with the wave's M tiles re-dealt so that one wave holds both x planes of a window (8 x 8 x 8 tile, wave =
// x pair x 4 y rows x 8 z) the other seven window values are the lane's own second tile (x), lane ^ 8 (y) and lane ^ 1 (z):
// window maximum and first-maximum index by two DPP steps each.  fp32x3 / x1, 64-column tile, even volume dims.
constexpr int CV_POOLF = 1 << 23;
// Folded upsampling.  A decoder layer convolves concat([UpSampling3D(2)(x) | skip]).  For the upsampled channels the 27
// taps at a full-resolution voxel g = 2 i + p touch only a 2x2x2 block of x: per axis, parity p = 0 reads x[i-1] with
// W[-1] and x[i] with W[0] + W[+1]; p = 1 reads x[i] with W[-1] + W[0] and x[i+1] with W[+1] (zero padding of the
// upsampled tensor = zero padding of x).  So that half of the layer is 8 convolutions (one per parity class) with 8 taps
// each ON THE LOW-RESOLUTION GRID: 64 tap-steps per low-res voxel instead of 8 x 27 = 216 -- 8/27 of the MACs.
//   CV_UPFOLD: the kernel runs on x (p.X/Y/Z = low-res dims), blockIdx.y = class * ntiles_n + n-tile, walks the class's
//              8 taps (halo offsets p + s, s in {0,1}: the same offsets a 3x3x3 tap has) over pre-folded weights
//              (mmr_conv3d_k3_upfold_pack) and stores the raw fp32 sums at (2 i + p) of a full-resolution tensor;
//   CV_CINIT:  the ordinary 27-tap kernel over the skip channels starts its accumulators from that tensor.
// dec_final_0 of BASELINE configs[1] (512 -> 256 at 160x160x192): 17.4 -> 11.3 TMAC.
constexpr int CV_UPFOLD = 1 << 16;
constexpr int CV_CINIT = 1 << 17;
// The partial tensor between the two launches as IEEE half (saturated to +-65504) instead of fp32 -- bf16 layers only: the
// partial's 2^-12 relative rounding disappears under the 2^-9 of the layer's bf16 output, and the 10 GB round trip of
// dec_final_0 at 160x160x192 x 256 (written by one launch, read by the next, neither overlapped with MFMA work at one
// workgroup per CU) halves.  fp32x3 layers keep the fp32 partial.
constexpr int CV_PART16 = 1 << 18;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
// The data gradient of the folded half (training): d x[j] = sum over parity classes p and tap bits s of
// Wf[p][s]^T dz[2 (j + 1 - p - s) + p] (per axis) -- again an 8-tap convolution per class on the low-resolution grid, now
// gathering from the class's sub-lattice of the full-resolution dz and summing all 8 classes into ONE accumulator: the
// K walk is (class, dz-channel slice) -> 8 taps, halo offset 2 - p - s.  64 tap-steps per low-res voxel instead of 216,
// the gradient lands compact at low resolution (no full-resolution intermediate, no 2x2x2 pooling pass), and the masked
// epilogue (LeakyReLU backward + bias sums of the layer that produced x) applies unchanged.  fp32x3 / x1 tensors only.
constexpr int CV_DGFOLD = 1 << 19;

#ifdef MMR_DIAG
int g_diag_stamps = 0;   // host switch of the stamped instantiations (conv and wgrad), set by mmr_debug_set_stamps
// Diagnostic build only (CV_STAMP): where a tap's cycles go.  Per wave slot w (0..7) the sums over all workgroups of:
// [0] tap top -> weight DMA issued, [1] -> last MFMA issued (fragment reads + MFMAs), [2] -> own DMA landed (vmcnt 0),
// [3] -> barrier passed, [4] A restage (per slice), [5] number of taps.  s_memtime ticks.  Read the SHARES, never the
// run time of this build (cdna_hip_programming.md section 7, in-kernel stamps).
__device__ unsigned long long g_conv_stamp[8][8];
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MMR_STAMP(k) do { if constexpr (STAMP) { const unsigned long long t_ = stamp_now(); st_acc[k] += t_ - st_t; st_t = t_; } } while (0)
#else
#define MMR_STAMP(k) do { } while (0)
#endif

template <int DT> struct Elt;
template <> struct Elt<MMR_DT_BF16> { static constexpr int size = 2; static constexpr int kc = 64; };
template <> struct Elt<MMR_DT_F32> { static constexpr int size = 4; static constexpr int kc = 32; };
// fp32 tensors in HBM, split on the fly into bf16 (hi, lo) pairs: a*b ~ hi*hi + hi*lo + lo*hi on the bf16 MFMA
// (3 MFMAs at 16x the fp32-MFMA rate, ~1e-5 relative error): LDS row = [32 ch hi | 32 ch lo] = 128 B.
template <> struct Elt<MMR_DT_F32X3> { static constexpr int size = 4; static constexpr int kc = 32; };
// fp32 tensors, products on the hi halves only (one bf16 MFMA, bf16-grade): opt-in for the backward pass
template <> struct Elt<MMR_DT_F32X1> { static constexpr int size = 4; static constexpr int kc = 32; };

// LDS-DMA (global_load_lds_dwordx4) issued through inline asm so that hipcc does NOT see an LDS write:
// with the builtin it cannot prove that the DMA destination (B buffer cur^1) and the fragment reads (sA,
// B buffer cur) are disjoint and puts `s_waitcnt vmcnt(0)` in front of the first ds_read of every tap,
// which serialises the weight prefetch with the MFMAs (measured: 1.1 -> see DESIGN.md).  The DMA is
// retired by the explicit vmcnt wait in front of the end-of-tap barrier.  `lds_off` = wave-uniform LDS
// byte address of lane 0's 16 bytes; lane i lands at lds_off + 16 i.
__device__ __forceinline__ void glds16(const char* g, unsigned lds_off)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_off)
                 : "memory");
}

// Same, with the source as a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: no 64-bit address VGPRs,
// so the DMA can be issued from the middle of the MFMA stream where every fragment register is live.
__device__ __forceinline__ void glds16_s(const char* sbase_in, unsigned voff, unsigned lds_off)
{
    // the "s" constraint does not make a value uniform: hand the compiler one that provably is.  The base then comes from
    // v_readfirstlane (a VALU write of an SGPR) and hipcc pads nothing inside an asm: a VMEM instruction that reads such an
    // SGPR needs 5 wait states, hence the `s_nop 4`
    const unsigned long long a = (unsigned long long)sbase_in;
    // (readfirstlane returns int: go through unsigned, or the low half is SIGN-extended over the high one)
    const unsigned long long sbase = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a) |
                                     ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_off)
                 : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const void* p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

// XOR swizzle of the 16-B chunk index inside a 128-B LDS row, keyed on the halo coordinates.
// A ds_read_b128 lane group covers 2 consecutive y x 8 consecutive z voxels (see row_perm), for which
// (hz & 1, swz) takes 16 distinct values -> every tap's fragment read is bank-conflict free with
// unpadded rows (rocprof before: SQ_LDS_BANK_CONFLICT = 56 % of SQ_LDS_IDX_ACTIVE with 144-B rows).
__device__ __forceinline__ int swz(int hy, int hz) { return ((hz >> 1) & 3) | ((hy & 1) << 2); }
// Swizzle of the 16x16x32 kernels.  Their operand lanes are (voxel r16 = 2 y-rows x 8 z, k-chunk q16), so a ds_read_b128
// lane group {0-3, 12-15, 20-27} holds (y0, z0-3) and (y1, z4-7) at chunk q and (y0, z4-7), (y1, z0-3) at chunk q ^ 1.  Rows
// are 128 B, i.e. half a 256-B bank row, the half being the parity of hz: per parity the group has 8 cells whose
// (hz >> 1) & 3 takes every value twice, once under each chunk parity, so chunk ^ (((hz >> 1) & 3) << 1) is a bijection onto
// the 8 slots.  The 32x32x16 swizzle above was still in use on this path and left 2/3 of the A reads 2-way conflicted
// (SQ_LDS_BANK_CONFLICT = 31 % of SQ_LDS_IDX_ACTIVE, profiles/r02a_infer_pmc_sq.json).
__device__ __forceinline__ int swz16(int hz) { return ((hz >> 1) & 3) << 1; }

// MFMA A-row r (= lane & 31) -> voxel y_local*8 + z inside the 4(y) x 8(z) patch of an M-tile, chosen so
// that the hardware's ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} each hold two
// whole y-rows.
__device__ __forceinline__ int row_perm(int r)
{
    const int q = r >> 2;  // 0..7
    // q: 0 -> +0, 1,2 -> +12, 3 -> -8, 4 -> +8, 5,6 -> -12, 7 -> +0
    const int d = (q == 0 || q == 7) ? 0 : (q <= 2 ? 12 : (q == 3 ? -8 : (q == 4 ? 8 : -12)));
    return r + d;
}

// lane ^ 1 (quad_perm [1,0,3,2]) and lane ^ 8 (row_ror:8 inside a row of 16) as DPP moves -- CV_POOLF's window partners
__device__ __forceinline__ int dpp_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_xor8(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true); }
__device__ __forceinline__ float dpp_xor1(float v) { return __int_as_float(dpp_xor1(__float_as_int(v))); }
__device__ __forceinline__ float dpp_xor8(float v) { return __int_as_float(dpp_xor8(__float_as_int(v))); }

template <int DT, int WM, int WN, int MT, int NT, int VAR>
__global__ void __launch_bounds__(CONV_THREADS, 2)
conv3d_k3_kernel(const ConvParams p)
{
    // output tile = TXT x 8 x 8 voxels (TXT = 4: 256 rows, TXT = 8: 512 rows for the narrow-N configurations,
    // which doubles the MFMA work between two barriers and halves the fragment reads per MFMA)
    constexpr int TXT = WM * MT * 32 / (TY * TZ);
    constexpr int HXT = TXT + 2;
    constexpr int HROWS_T = HXT * HY * HZ;
    constexpr int A_BYTES_T = HROWS_T * ROWB;
    // CV_M16: v_mfma_f32_16x16x32_bf16 instead of 32x32x16: same bytes per flop, the chip holds a higher clock on
    // this shape (MI355X_MICROARCH.md, DVFS give-back item 7)
    constexpr bool M16 = ((VAR & CV_M16) != 0) && (DT == MMR_DT_BF16 || DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1);
    constexpr bool LO = (DT == MMR_DT_F32X3);  // lo halves staged and multiplied
    // CV_PIPE, bf16 128x64 wave tile: explicit fragment pipeline, and the weight DMA of the next tap is issued from
    // INSIDE the MFMA stream by waves 0-3 only (wave w shares a SIMD with wave w + 4 and wins the issue arbitration by
    // age): their own pieces after MFMA group 4 of 16, the pieces of wave w + 4 after group 10; waves 4-7 never issue.
    // Issued at the top of the tap by everyone, the DMA (and the fragment round trip behind it) left the matrix pipe
    // idle for 220-340 clk right after every barrier (cycle stamps).
    constexpr bool MIDDMA = M16 && (DT == MMR_DT_BF16) && MT == 4 && ((VAR & CV_PIPE) != 0);
    constexpr bool PIPE3 = M16 && (DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1) && ((VAR & CV_PIPE) != 0);
    constexpr bool BATCHA = (DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1) && ((VAR & CV_BATCHA) != 0);
    // static priority for the younger half of the workgroup (-1 % on the 64-column fp32x3 tile, +14 % on the 256-column
    // bf16 tile whose DMA issue needs waves 0-3 to be the arbitration winners)
    constexpr bool PRIO_Y = (VAR & CV_PRIO_Y) != 0;
    constexpr bool POOLF = (VAR & CV_POOLF) != 0;
    static_assert(!POOLF || (((VAR & CV_M16) != 0) && WM == 8 && MT == 2 && NT == 2 && DT != MMR_DT_BF16 && DT != MMR_DT_F32),
                  "pooling-gradient epilogue: fp32x3 / x1, 64-column tile");
    constexpr bool UPF = (VAR & CV_UPFOLD) != 0;
    constexpr bool CINIT = (VAR & CV_CINIT) != 0;
    constexpr bool PART16 = (VAR & CV_PART16) != 0;
    constexpr bool DGF = (VAR & CV_DGFOLD) != 0;
    static_assert(!DGF || (!UPF && !CINIT && ((VAR & CV_M16) != 0) && ((VAR & CV_BATCHA) != 0) &&
                           (DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1)), "dgrad fold: fp32x3 / x1, batched staging");
    constexpr int TAPS = (UPF || ((VAR & CV_DGFOLD) != 0)) ? 8 : 27;
    static_assert(!(UPF || CINIT) || (((VAR & CV_M16) != 0) && DT != MMR_DT_F32 && NT == 2), "folded upsampling: 16x16x32 kernels");
#ifdef MMR_DIAG
    constexpr bool STAMP = (VAR & CV_STAMP) != 0;
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = 0;
#endif
    static_assert(WM * WN == 8, "8 waves");
    static_assert(TXT == 4 || TXT == 8, "M tile");
    constexpr int BN = WN * NT * 32;
    constexpr int ES = Elt<DT>::size;
    constexpr int KC = Elt<DT>::kc;
    constexpr int B_BYTES = BN * 128;
    constexpr int B_ITERS = B_BYTES / (CONV_THREADS * 16);
    constexpr int A_ITERS = (((DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1) ? HROWS_T * 4 : HROWS_T * 8) + CONV_THREADS - 1) / CONV_THREADS;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + A_BYTES_T;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5;

    const int ntn = UPF ? (p.Cout + BN - 1) / BN : 1;
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if constexpr (UPF) {
        // Which (tile, class) a workgroup takes.  Dispatch order is x fastest and workgroup L lands on XCD L & 7, so with
        // blockIdx.y = class every XCD works through ONE class at a time and each of the 8 classes fetches the layer's whole input
        // again from HBM / the Infinity Cache (C2: 6.3 GB per step for 0.36 GB of input).  xcd_pair = P: a group of P XCDs takes P
        // classes of all tiles, a tile's P classes back to back on ONE XCD -- all but the first find the haloed A tile in that
        // XCD's L2, which then holds P classes' weights instead of one.  A bijection for every grid size (checked on the host for
        // P = 2, 4, 8).  Same-box, same library, off | on (profiles/r05_ab_upfold_xcd_groups.log): bf16 dec_final_0 -3.3 / -2.8 /
        // -2.8 % for P = 2 / 4 / 8, dec_conv_3 -1.1 / -1.3 / +0.7 %, fp32x3 dec_final_0 -0.6 / -3.6 / -2.8 %: P = 4.
        if (p.xcd_pair) {
            const int P = p.xcd_pair;                              // classes per XCD group
            const int n = (int)gridDim.x * ntn;                    // (tile, n-tile) blocks per class = workgroups per XCD
            const int L = by * (int)gridDim.x + bx, x = L & 7, k = L >> 3;
            const int xg = x % P, full = (n / P) * P, r = n - full;
            int g, c;
            if (k < full) { g = (k / P) * P + xg; c = k % P; }
            else { const int t = xg * r + (k - full); g = full + t / P; c = t % P; }
            bx = g / ntn;
            by = ((x / P) * P + c) * ntn + g % ntn;
        }
    }
    int bid = bx + p.tile0;
    const int tzi = bid % p.ntz; bid /= p.ntz;
    const int tyi = bid % p.nty; bid /= p.nty;
    const int txi = bid % p.ntx;
    const int b = bid / p.ntx;
    const int x0 = txi * TXT, y0 = tyi * TY, z0 = tzi * TZ;
    const int cls = UPF ? by / ntn : 0;                 // parity class (px, py, pz) = bits 2, 1, 0
    const int ntile = UPF ? by % ntn : by;

    const int pv = row_perm(lane & 31);
    const int vyl = pv >> 3, vz = pv & 7;  // y inside the M-tile's 4-row patch, z
    int a_off[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mt = wm * MT + m;  // M-tile index: x = mt >> 1, y base = (mt & 1) * 4
        a_off[m] = (((mt >> 1) * HY + (mt & 1) * 4 + vyl) * HZ + vz) * ROWB;
    }
    int b_off[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) b_off[n] = ((h * BN) + (wn * NT + n) * 32 + (lane & 31)) * 16;

    f32x16 acc[M16 ? 1 : MT][M16 ? 1 : NT];
    if constexpr (!M16) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    }
    // 16x16x32 path: 16-voxel tiles = 2 y-rows x 8 z; operand lane l: voxel (or cout) l & 15, k-chunk l >> 4.
    // The product is formed TRANSPOSED (weights as the MFMA's A operand, voxels as B): the C/D layout then gives
    // lane (r16, q16) the four consecutive couts q16*4 .. +3 of voxel r16 -> 8-B (bf16) / 16-B (fp32) stores.
    f32x4 acc16[M16 ? 2 * MT : 1][M16 ? 2 * NT : 1];
    int a16_off[M16 ? 2 * MT : 1], b16_off[M16 ? 2 * NT : 1];
    const int r16 = lane & 15, q16 = lane >> 4;
    // 16-voxel tile mi of this wave = 2 y rows x 8 z at (x, y) = (t16x(mi), t16y(mi)) of the workgroup's tile.  Ordinarily a wave
    // owns MT consecutive 32-row M tiles; CV_POOLF deals them so that a wave holds BOTH x planes of a pooling window
    auto t16x = [&](int mi) { return POOLF ? 2 * (wm >> 1) + (mi >> 1) : (wm * MT + (mi >> 1)) >> 1; };
    auto t16y = [&](int mi) { return POOLF ? (wm & 1) * 4 + 2 * (mi & 1) : ((wm * MT + (mi >> 1)) & 1) * 4 + 2 * (mi & 1); };
    if constexpr (M16) {
#pragma unroll
        for (int mi = 0; mi < 2 * MT; ++mi) {
            a16_off[mi] = ((t16x(mi) * HY + t16y(mi) + (r16 >> 3)) * HZ + (r16 & 7)) * ROWB;
#pragma unroll
            for (int ni = 0; ni < 2 * NT; ++ni) acc16[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ni = 0; ni < 2 * NT; ++ni) b16_off[ni] = (q16 * BN + wn * NT * 32 + ni * 16 + r16) * 16;
        if constexpr (CINIT) {
            // accumulator (mi, ni)[r] <-> cout co + ni * 4 + r of voxel (mi, r16): the epilogue's layout, read the same way.
            // Branch-free: every lane loads from a clamped, always-valid address and out-of-tile values are zeroed afterwards
            // -- behind `if (voxel in volume)` hipcc waits for each tile's loads before it issues the next tile's (eight
            // dependent round trips per workgroup: 14 us of a 168-us tile, +9 % on the 256 -> 256 layer at 160x160x192;
            // branch-free +5.7 %).  Adding the partial in the EPILOGUE instead (all loads in flight at its top) measured
            // slower still, 12.87 vs 12.23 ms for that layer: the prologue's loads at least share the wait for the A tile.
            const int co = ntile * BN + wn * 64 + q16 * 16;
            const int cco = co + 15 < p.Cout ? co : 0;
            bool okv[2 * MT];
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            u32x4 raw16[2 * MT][2];
#pragma unroll
            for (int mi = 0; mi < 2 * MT; ++mi) {
                const int mt = wm * MT + (mi >> 1);
                const int gx = x0 + (mt >> 1), gy = y0 + (mt & 1) * 4 + 2 * (mi & 1) + (r16 >> 3), gz = z0 + (r16 & 7);
                okv[mi] = co + 15 < p.Cout && gx < p.X && gy < p.Y && gz < p.Z && (!p.kpart || blockIdx.z == 0);   // split K: block 0 carries the init
                const int cx = gx < p.X ? gx : p.X - 1, cy = gy < p.Y ? gy : p.Y - 1, cz = gz < p.Z ? gz : p.Z - 1;
                const size_t o = ((((size_t)b * p.X + cx) * p.Y + cy) * p.Z + cz) * p.Cout + cco;
                if constexpr (PART16) {
                    const _Float16* ci = reinterpret_cast<const _Float16*>(p.cinit) + o;
                    raw16[mi][0] = *reinterpret_cast<const u32x4*>(ci);
                    raw16[mi][1] = *reinterpret_cast<const u32x4*>(ci + 8);
                } else {
                    const float* ci = p.cinit + o;
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc16[mi][ni] = *reinterpret_cast<const f32x4*>(ci + ni * 4);
                }
            }
#pragma unroll
            for (int mi = 0; mi < 2 * MT; ++mi) {
                if constexpr (PART16) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const f16x8 hv = __builtin_bit_cast(f16x8, raw16[mi][k]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc16[mi][2 * k + (e >> 2)][e & 3] = okv[mi] ? (float)hv[e] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc16[mi][ni][r] = okv[mi] ? acc16[mi][ni][r] : 0.f;
                }
            }
        }
    }

    const int ncs = (p.C0 + p.C1) / KC;              // channel slices of the input
    const int nslices = DGF ? 8 * ncs : ncs;         // dgrad fold: (parity class, dz-channel slice) pairs
    const int G = nslices * TAPS;
    const char* wtile = p.wp + (size_t)by * G * B_BYTES;   // UPF: [class][n-tile], else [n-tile]
    const int X2 = p.X >> 1, Y2 = p.Y >> 1, Z2 = p.Z >> 1;

    // global load of item `it` of this thread's share of the haloed tile of channel slice s.
    // bf16 / fp32: item = (row, 16-B chunk); fp32x3: item = (row, 8-channel group) = 32 B of fp32.
    constexpr bool X3 = (DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1);  // fp32 in HBM, [hi | lo] LDS rows
    constexpr int A_ITEMS = X3 ? HROWS_T * 4 : HROWS_T * 8;
    struct AItem { uint4 a, b; };
    auto load_a = [&](int s, int it) -> AItem {
        // (see dma_stage_a; on this register-staging path the recomputation costs more than the registers it frees -- fp32x3 C2
        // forward 93.4 -> 99.0 ms, the 256-column _cinit 92.2 -> 95.4 even though it spills -- except for the 64-column _cinit
        // tile of the training step: 2.75 -> 2.65 ms)
        constexpr bool UNHOIST = CINIT && BN == 64;
        int tid_l = tid;
        if constexpr (UNHOIST) asm volatile("" : "+v"(tid_l));
        const int i = tid_l + it * CONV_THREADS;
        AItem val;
        val.a = make_uint4(0, 0, 0, 0);
        val.b = make_uint4(0, 0, 0, 0);
        if (i < A_ITEMS) {
            const int ch0 = s * KC;
            const bool first = ch0 < p.C0;
            const char* src = first ? p.in0 : p.in1;
            const int Cs = first ? p.C0 : p.C1;
            const int chs = first ? ch0 : ch0 - p.C0;
            const bool up = first && p.up0;
            const int row = X3 ? (i >> 2) : (i >> 3), chunk = X3 ? (i & 3) : (i & 7);
            const int hx = row / (HY * HZ), hy = (row / HZ) % HY, hz = row % HZ;
            const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
            if (gx >= 0 && gx < p.X && gy >= 0 && gy < p.Y && gz >= 0 && gz < p.Z) {
                size_t vox;
                if (up) vox = (((size_t)b * X2 + (gx >> 1)) * Y2 + (gy >> 1)) * Z2 + (gz >> 1);
                else vox = (((size_t)b * p.X + gx) * p.Y + gy) * p.Z + gz;
                const char* q = src + (vox * Cs + chs) * ES + chunk * (X3 ? 32 : 16);
                val.a = *reinterpret_cast<const uint4*>(q);
                if constexpr (X3) val.b = *reinterpret_cast<const uint4*>(q + 16);
            }
        }
        return val;
    };
    // CV_BATCHA (fp32x3 / x1): the same items, but every lane loads from a clamped, always-valid address and returns the
    // in-bounds flag separately, so that all loads of a slice can be issued back to back and masked when they are stored.
    // `if (in bounds) load` compiles to a branch + wait per item: the 8 items of a slice were 8 dependent round trips
    // (6.7 k cycles per slice on the stamps of tools/conv_stamps.py train).
    struct AItemM { uint4 a, b; unsigned ok; };
    auto load_a_nb = [&](int s, int it) -> AItemM {
        int tid_l = tid;
        if constexpr (CINIT && BN == 64) asm volatile("" : "+v"(tid_l));   // see load_a
        const int i0 = tid_l + it * CONV_THREADS;
        const int i = i0 < A_ITEMS ? i0 : A_ITEMS - 1;
        const int scls = DGF ? s / ncs : 0;                     // dgrad fold: parity class of this K block
        const int ch0 = (DGF ? s - scls * ncs : s) * KC;
        const bool first = ch0 < p.C0;
        const char* src = first ? p.in0 : p.in1;
        const int Cs = first ? p.C0 : p.C1;
        const int chs = first ? ch0 : ch0 - p.C0;
        const bool up = first && p.up0;
        const int row = i >> 2, chunk = i & 3;
        const int hx = row / (HY * HZ), hy = (row / HZ) % HY, hz = row % HZ;
        const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
        const unsigned ok = (i0 < A_ITEMS && gx >= 0 && gx < p.X && gy >= 0 && gy < p.Y && gz >= 0 && gz < p.Z) ? 0xffffffffu : 0u;
        const int pX = p.X, pY = p.Y, pZ = p.Z;
        const int cx = gx < 0 ? 0 : (gx >= pX ? pX - 1 : gx), cy = gy < 0 ? 0 : (gy >= pY ? pY - 1 : gy),
                  cz = gz < 0 ? 0 : (gz >= pZ ? pZ - 1 : gz);
        size_t vox;
        if constexpr (DGF)   // row = low-res voxel i of the class's sub-lattice of the full-resolution tensor: voxel 2 i + p
            vox = (((size_t)b * (2 * pX) + 2 * cx + ((scls >> 2) & 1)) * (2 * pY) + 2 * cy + ((scls >> 1) & 1)) * (2 * pZ) +
                  2 * cz + (scls & 1);
        else if (up) vox = (((size_t)b * X2 + (cx >> 1)) * Y2 + (cy >> 1)) * Z2 + (cz >> 1);
        else vox = (((size_t)b * pX + cx) * pY + cy) * pZ + cz;
        const char* q = src + (vox * Cs + chs) * ES + chunk * 32;
        AItemM val;
        val.a = *reinterpret_cast<const uint4*>(q);
        val.b = *reinterpret_cast<const uint4*>(q + 16);
        val.ok = ok;
        return val;
    };
    auto and4 = [](uint4 v, unsigned m) { return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m); };
    auto store_a = [&](int it, const AItem& val) {
        int tid_l = tid;
        if constexpr (CINIT && BN == 64) asm volatile("" : "+v"(tid_l));   // see load_a
        const int i = tid_l + it * CONV_THREADS;
        if (i < A_ITEMS) {
            const int row = X3 ? (i >> 2) : (i >> 3), chunk = X3 ? (i & 3) : (i & 7);
            const int hy = (row / HZ) % HY, hz = row % HZ;
            const int sz_ = M16 ? swz16(hz) : swz(hy, hz);
            if constexpr (X3) {
                const unsigned u[8] = {val.a.x, val.a.y, val.a.z, val.a.w, val.b.x, val.b.y, val.b.z, val.b.w};
                unsigned hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f0 = __uint_as_float(u[2 * e]), f1 = __uint_as_float(u[2 * e + 1]);
                    const bf16_t h0 = f32_to_bf16(f0), h1 = f32_to_bf16(f1);
                    const bf16_t l0 = f32_to_bf16(f0 - bf16_to_f32(h0)), l1 = f32_to_bf16(f1 - bf16_to_f32(h1));
                    hi[e] = (unsigned)h0 | ((unsigned)h1 << 16);
                    lo[e] = (unsigned)l0 | ((unsigned)l1 << 16);
                }
                *reinterpret_cast<uint4*>(sA + row * ROWB + ((chunk ^ sz_) << 4)) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                if constexpr (LO)
                    *reinterpret_cast<uint4*>(sA + row * ROWB + (((chunk + 4) ^ sz_) << 4)) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            } else {
                *reinterpret_cast<uint4*>(sA + row * ROWB + ((chunk ^ sz_) << 4)) = val.a;
            }
        }
    };
    const unsigned sB_lds = lds_addr(sB);
    const unsigned tid16 = (unsigned)tid * 16u;
    auto issue_b = [&](int g, int buf, int wv = -1) {   // wv: issue the pieces of wave slot wv (default: this wave's own)
        const char* wt = wtile + (size_t)g * B_BYTES;
        const int ws_ = wv < 0 ? wave : wv;
        const unsigned dst = __builtin_amdgcn_readfirstlane(sB_lds + buf * B_BYTES + ws_ * 1024);
        const unsigned t16 = (unsigned)((ws_ << 6) | lane) * 16u;
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            if constexpr (MIDDMA) glds16_s(wt + it * CONV_THREADS * 16, wv < 0 ? tid16 : t16, dst + it * CONV_THREADS * 16);
            else glds16(wt + (it * CONV_THREADS + tid) * 16, dst + it * CONV_THREADS * 16);
        }
        if (B_BYTES < CONV_THREADS * 16) {
            if (wave * 1024 < B_BYTES) glds16(wt + tid * 16, dst);
        }
    };

    // CV_DMA_A (bf16 / exact fp32, no conversion on the way): the haloed A tile goes global -> LDS by DMA as well.
    // Lane i of an instruction lands at base + 16 i, i.e. at (row, chunk position) = (i >> 3, i & 7) of the swizzled
    // tile, so the swizzle is applied to the SOURCE chunk; out-of-volume rows read a zero page.
    constexpr bool DMA_A = !X3 && ((VAR & CV_DMA_A) != 0);
    const unsigned sA_lds = lds_addr(sA);
    constexpr bool ATAB = DMA_A && MIDDMA && !DGF && TXT == 4;
    unsigned* atab = reinterpret_cast<unsigned*>(sB + 2 * B_BYTES);      // [A_ITEMS] item -> source byte offset, ~0 = zero row
    bool atab_ok = false;
    const char* atab_base = nullptr;
    if constexpr (ATAB) {
        atab_ok = p.C1 == 0 && !p.up0 && (long long)HXT * p.Y * p.Z * p.C0 * ES < (1ll << 32) - 128;
        issue_b(p.kpart ? (int)blockIdx.z * p.gsplit : 0, 0);   // the first weight block is under way before the table arithmetic
        if (atab_ok) {
            const int ox = x0 > 0 ? x0 - 1 : 0, oy = y0 > 0 ? y0 - 1 : 0, oz = z0 > 0 ? z0 - 1 : 0;   // first in-volume halo voxel
            atab_base = p.in0 + ((((size_t)b * p.X + ox) * p.Y + oy) * p.Z + oz) * (size_t)p.C0 * ES;
            const unsigned rs = (unsigned)p.C0 * ES;
            // the first slice of this block's K range goes out from the registers while the table is being written: its DMAs
            // overlap the arithmetic of the items behind them instead of waiting for the whole table (1.6 k cycles per tile)
            const int s_first = (p.kpart ? (int)blockIdx.z * p.gsplit : 0) / TAPS;
            const char* sb_first = atab_base + (size_t)s_first * (KC * ES);
#pragma unroll
            for (int it = 0; it < A_ITERS; ++it) {
                const int i = tid + it * CONV_THREADS;
                if (i < A_ITEMS) {
                    const int row = i >> 3, cpos = i & 7;
                    const int hx = row / (HY * HZ), hy = (row / HZ) % HY, hz = row % HZ;
                    const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
                    const bool ok = gx >= 0 && gx < p.X && gy >= 0 && gy < p.Y && gz >= 0 && gz < p.Z;
                    const unsigned rel = (unsigned)(((gx - ox) * p.Y + (gy - oy)) * p.Z + (gz - oz));
                    const unsigned e = rel * rs + (unsigned)((cpos ^ (M16 ? swz16(hz) : swz(hy, hz))) << 4);
                    atab[i] = ok ? e : 0xffffffffu;
                    if (ok) {
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep)
                                     : "v"(e), "s"(sb_first), "s"(sA_lds + (it * CONV_THREADS + wave * 64) * 16)
                                     : "memory");
                    } else {
                        *reinterpret_cast<uint4*>(sA + i * 16) = make_uint4(0, 0, 0, 0);
                    }
                }
            }
        }
    }
    auto dma_stage_a = [&](int s) {
        // The row -> address arithmetic below is invariant across slices and hipcc hoists it out of the tap loop: ~60 VGPRs per
        // lane stay live through the whole K walk.  The plain kernel just fits (256 VGPRs, 8 B of scratch); the _cinit
        // instantiation, whose prologue holds the accumulator-init loads on top, spilled 100 B per lane and paid 25 scratch
        // reloads in EVERY restage -- most of what made it 5.7 % slower than the plain kernel.  Keeping the arithmetic inside
        // the restage instead (an opaque copy of the thread index) costs the plain kernel 2 % and the folded launch 3 %.
        // ATAB (bf16 256-column tile, one directly-read input): the byte offset of every (row, chunk) item from the tile
        // origin is computed ONCE per workgroup into the 19 KB of LDS the tile leaves free; a restage is then one ds_read and
        // one DMA per item -- no arithmetic, nothing hoisted (~200 VGPRs, no scratch).  Rows outside the volume are zeroed once
        // and left out of the DMA.
        if constexpr (ATAB) {
            if (atab_ok) {
                const char* sb = atab_base + (size_t)s * (KC * ES);
#pragma unroll
                for (int it = 0; it < A_ITERS; ++it) {
                    const int i = tid + it * CONV_THREADS;
                    if (i < A_ITEMS) {
                        const unsigned e = atab[i];
                        if (e != 0xffffffffu) {
                            unsigned keep;
                            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                         : "=&s"(keep)
                                         : "v"(e), "s"(sb), "s"(sA_lds + (it * CONV_THREADS + wave * 64) * 16)
                                         : "memory");
                        }
                    }
                }
                return;
            }
        }
        // general form (two inputs, nearest-upsampled first input): addresses from scratch.  In the table kernels an opaque
        // copy of the thread index keeps this arithmetic from being hoisted over the tap loop all the same.
        int tid_g = tid;
        if constexpr (ATAB) asm volatile("" : "+v"(tid_g));
        const int ch0 = s * KC;
        const bool first = ch0 < p.C0;
        const char* src = first ? p.in0 : p.in1;
        const int Cs = first ? p.C0 : p.C1;
        const int chs = first ? ch0 : ch0 - p.C0;
        const bool up = first && p.up0;
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
            const int i = tid_g + it * CONV_THREADS;
            if (i < A_ITEMS) {
                const int row = i >> 3, cpos = i & 7;
                const int hx = row / (HY * HZ), hy = (row / HZ) % HY, hz = row % HZ;
                const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
                const char* q = reinterpret_cast<const char*>(&g_zero16);
                if (gx >= 0 && gx < p.X && gy >= 0 && gy < p.Y && gz >= 0 && gz < p.Z) {
                    size_t vox;
                    if (up) vox = (((size_t)b * X2 + (gx >> 1)) * Y2 + (gy >> 1)) * Z2 + (gz >> 1);
                    else vox = (((size_t)b * p.X + gx) * p.Y + gy) * p.Z + gz;
                    q = src + (vox * Cs + chs) * ES + ((cpos ^ (M16 ? swz16(hz) : swz(hy, hz))) << 4);
                }
                glds16(q, __builtin_amdgcn_readfirstlane(sA_lds + (it * CONV_THREADS + wave * 64) * 16));
            }
        }
    };

    // the epilogue's 16 bias values per lane, loaded here in the table kernels (they have the registers): their round trip
    // is then not the first thing a finished tile waits for
    constexpr bool BIAS_PRE = !UPF && !DGF && (ATAB || (X3 && BN == 64 && M16));
    float bias_pre[16];
    if constexpr (BIAS_PRE) {
        const int cop = (int)blockIdx.y * BN + wn * 64 + q16 * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) bias_pre[e] = (p.bias && cop + e < p.Cout) ? p.bias[cop + e] : 0.f;
    }
    // ---- prologue: first slice of A, first tap of B of this block's step range ----
    const int g0 = p.kpart ? (int)blockIdx.z * p.gsplit : 0;
    const int g1 = p.kpart ? min(g0 + p.gsplit, G) : G;
    int cur = 0, tap = g0 % TAPS, s = g0 / TAPS;
    if constexpr (!ATAB) issue_b(g0, 0);
    // a macro, not a lambda: wrapped in one more closure, hipcc no longer scalarises the by-value kernel argument
    // struct and every instantiation reads ConvParams from scratch (288 B/lane; the bn64 convs ran 1.8x slower)
    // named scalars, not an array: an indexed array of structs stays in scratch here even when unrolled
#define MMR_STAGE_A_REGS(sl) \
    do { \
        if constexpr (BATCHA) { \
            static_assert(A_ITERS <= 10, "staging items"); \
            AItemM i0; \
            AItemM i1; \
            AItemM i2; \
            AItemM i3; \
            AItemM i4; \
            AItemM i5; \
            AItemM i6; \
            AItemM i7; \
            AItemM i8; \
            AItemM i9; \
            if constexpr (A_ITERS > 0) i0 = load_a_nb(sl, 0); \
            if constexpr (A_ITERS > 1) i1 = load_a_nb(sl, 1); \
            if constexpr (A_ITERS > 2) i2 = load_a_nb(sl, 2); \
            if constexpr (A_ITERS > 3) i3 = load_a_nb(sl, 3); \
            if constexpr (A_ITERS > 4) i4 = load_a_nb(sl, 4); \
            if constexpr (A_ITERS > 5) i5 = load_a_nb(sl, 5); \
            if constexpr (A_ITERS > 6) i6 = load_a_nb(sl, 6); \
            if constexpr (A_ITERS > 7) i7 = load_a_nb(sl, 7); \
            if constexpr (A_ITERS > 8) i8 = load_a_nb(sl, 8); \
            if constexpr (A_ITERS > 9) i9 = load_a_nb(sl, 9); \
            if constexpr (A_ITERS > 0) { AItem v; v.a = and4(i0.a, i0.ok); v.b = and4(i0.b, i0.ok); store_a(0, v); } \
            if constexpr (A_ITERS > 1) { AItem v; v.a = and4(i1.a, i1.ok); v.b = and4(i1.b, i1.ok); store_a(1, v); } \
            if constexpr (A_ITERS > 2) { AItem v; v.a = and4(i2.a, i2.ok); v.b = and4(i2.b, i2.ok); store_a(2, v); } \
            if constexpr (A_ITERS > 3) { AItem v; v.a = and4(i3.a, i3.ok); v.b = and4(i3.b, i3.ok); store_a(3, v); } \
            if constexpr (A_ITERS > 4) { AItem v; v.a = and4(i4.a, i4.ok); v.b = and4(i4.b, i4.ok); store_a(4, v); } \
            if constexpr (A_ITERS > 5) { AItem v; v.a = and4(i5.a, i5.ok); v.b = and4(i5.b, i5.ok); store_a(5, v); } \
            if constexpr (A_ITERS > 6) { AItem v; v.a = and4(i6.a, i6.ok); v.b = and4(i6.b, i6.ok); store_a(6, v); } \
            if constexpr (A_ITERS > 7) { AItem v; v.a = and4(i7.a, i7.ok); v.b = and4(i7.b, i7.ok); store_a(7, v); } \
            if constexpr (A_ITERS > 8) { AItem v; v.a = and4(i8.a, i8.ok); v.b = and4(i8.b, i8.ok); store_a(8, v); } \
            if constexpr (A_ITERS > 9) { AItem v; v.a = and4(i9.a, i9.ok); v.b = and4(i9.b, i9.ok); store_a(9, v); } \
        } else { \
            _Pragma("unroll") for (int it = 0; it < A_ITERS; ++it) store_a(it, load_a(sl, it)); \
        } \
    } while (0)
    if constexpr (DMA_A) {
        if (!(ATAB && atab_ok)) dma_stage_a(s);   // table kernels: already issued while the table was written
    } else {
        MMR_STAGE_A_REGS(s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (PRIO_Y) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }

    for (int g = g0; g < g1; ++g) {
        const bool more = g + 1 < g1;
#ifdef MMR_DIAG
        if constexpr (STAMP) st_t = stamp_now();
#endif
        if constexpr (!MIDDMA) {
            if (more) issue_b(g + 1, cur ^ 1);
        }
        MMR_STAMP(0);
        // halo offset of this tap; folded upsampling: parity bit + tap bit per axis
        const int kcls = DGF ? s / ncs : cls;     // dgrad fold: the class changes with the K block
        const int dx = UPF ? ((cls >> 2) & 1) + ((tap >> 2) & 1) : DGF ? 2 - ((kcls >> 2) & 1) - ((tap >> 2) & 1) : tap / 9;
        const int dy = UPF ? ((cls >> 1) & 1) + ((tap >> 1) & 1) : DGF ? 2 - ((kcls >> 1) & 1) - ((tap >> 1) & 1) : (tap / 3) % 3;
        const int dz = UPF ? (cls & 1) + (tap & 1) : DGF ? 2 - (kcls & 1) - (tap & 1) : tap % 3;
        const int tapoff = (dx * (HY * HZ) + dy * HZ + dz) * ROWB;
        const int sw = (swz(vyl + dy, vz + dz) ^ h) << 4;
        const char* bA = sA + tapoff;
        const char* bB = sB + cur * B_BYTES;
        if constexpr (M16) {
            const int sw16 = swz16((r16 & 7) + dz);
            if constexpr (X3 && PIPE3) {
                // fp32x3 / x1 with an explicit fragment pipeline: all B fragments and two A tiles up front, then per A
                // tile its MFMAs (lo*hi terms first, the dependent accumulations 4 apart) while the A tile two ahead loads
                static_assert(NT == 2 && (MT == 2 || MT == 4), "pipelined fp32x3 schedule: 64 columns per wave");
                uint4 ah16[2 * MT], al16[2 * MT], bh16[4], bl16[4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) bh16[ni] = *reinterpret_cast<const uint4*>(bB + b16_off[ni]);
                if constexpr (LO) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) bl16[ni] = *reinterpret_cast<const uint4*>(bB + b16_off[ni] + 4 * BN * 16);
                }
#pragma unroll
                for (int mi = 0; mi < 2 * MT; ++mi) {
                    ah16[mi] = *reinterpret_cast<const uint4*>(bA + a16_off[mi] + ((q16 ^ sw16) << 4));
                    if constexpr (LO) al16[mi] = *reinterpret_cast<const uint4*>(bA + a16_off[mi] + (((4 + q16) ^ sw16) << 4));
                }
#pragma unroll
                for (int mi = 0; mi < 2 * MT; ++mi) {
                    if constexpr (LO) {
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            acc16[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, bh16[ni]), __builtin_bit_cast(bf16x8, al16[mi]), acc16[mi][ni], 0, 0, 0);
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            acc16[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, bl16[ni]), __builtin_bit_cast(bf16x8, ah16[mi]), acc16[mi][ni], 0, 0, 0);
                    }
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc16[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, bh16[ni]), __builtin_bit_cast(bf16x8, ah16[mi]), acc16[mi][ni], 0, 0, 0);
                }
                constexpr int NA = LO ? 2 : 1, NM = LO ? 12 : 4;
#define MMR_GRP3(rd) __builtin_amdgcn_sched_group_barrier(0x008, NM, 0); if (rd) __builtin_amdgcn_sched_group_barrier(0x100, NA, 0)
                __builtin_amdgcn_sched_group_barrier(0x100, 4 * NA + 2 * NA, 0);
                if constexpr (MT == 4) { MMR_GRP3(1); MMR_GRP3(1); MMR_GRP3(1); MMR_GRP3(1); }
                MMR_GRP3(1); MMR_GRP3(1); MMR_GRP3(0); MMR_GRP3(0);
#undef MMR_GRP3
            } else if constexpr (MIDDMA) {
                static_assert(MT == 4 && NT == 2, "pipelined schedule is written for the 128x64 wave tile");
                uint4 fa[2][8], fb[2][4];
                const char* pa0 = bA + a16_off[0] + ((q16 ^ sw16) << 4);
                const char* pa1 = bA + a16_off[0] + (((4 + q16) ^ sw16) << 4);
                const char* pb = bB + b16_off[0];
                // fragment i in consumption order: B of k-step 0 (4), A of k-step 0 (8), B of k-step 1 (4), A of k-step 1 (8)
                auto rd = [&](int i) {
                    if (i < 4) fb[0][i] = *reinterpret_cast<const uint4*>(pb + i * 256);
                    else if (i < 12) { const int mi = i - 4; fa[0][mi] = *reinterpret_cast<const uint4*>(pa0 + (((mi >> 2) * HY + ((mi >> 1) & 1) * 4 + 2 * (mi & 1)) * HZ) * ROWB); }
                    else if (i < 16) { const int ni = i - 12; fb[1][ni] = *reinterpret_cast<const uint4*>(pb + ni * 256 + 4 * BN * 16); }
                    else { const int mi = i - 16; fa[1][mi] = *reinterpret_cast<const uint4*>(pa1 + (((mi >> 2) * HY + ((mi >> 1) & 1) * 4 + 2 * (mi & 1)) * HZ) * ROWB); }
                };
                auto mm = [&](int j) {   // MFMA j: k-step j / 32, A tile (j / 4) % 8, B fragment j % 4
                    const int ks = j >> 5, mi = (j >> 2) & 7, ni = j & 3;
                    acc16[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fb[ks][ni]), __builtin_bit_cast(bf16x8, fa[ks][mi]), acc16[mi][ni], 0, 0, 0);
                };
                // One code path, three scheduling regions cut by the two DMA issue points (the asm is a scheduling barrier):
                // MFMA groups 0..3 | first issue point | groups 4..9 | second issue point | groups 10..15.  Reads
                // go 7 up front, then per group of 4 MFMAs the counts below, split where the regions are cut.
#pragma unroll
                for (int i = 0; i < 12; ++i) rd(i);
#pragma unroll
                for (int j = 0; j < 16; ++j) mm(j);
                __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (more && wave < 4) issue_b(g + 1, cur ^ 1);
#pragma unroll
                for (int i = 12; i < 21; ++i) rd(i);
#pragma unroll
                for (int j = 16; j < 40; ++j) mm(j);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (more && wave < 4) issue_b(g + 1, cur ^ 1, wave + 4);
#pragma unroll
                for (int i = 21; i < 24; ++i) rd(i);
#pragma unroll
                for (int j = 40; j < 64; ++j) mm(j);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    uint4 fa16[2 * MT], fb16[2 * NT];
#pragma unroll
                    for (int mi = 0; mi < 2 * MT; ++mi)
                        fa16[mi] = *reinterpret_cast<const uint4*>(bA + a16_off[mi] + (((4 * ks + q16) ^ sw16) << 4));
#pragma unroll
                    for (int ni = 0; ni < 2 * NT; ++ni)
                        fb16[ni] = *reinterpret_cast<const uint4*>(bB + b16_off[ni] + ks * 4 * BN * 16);
#pragma unroll
                    for (int mi = 0; mi < 2 * MT; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2 * NT; ++ni)
                            acc16[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, fb16[ni]), __builtin_bit_cast(bf16x8, fa16[mi]), acc16[mi][ni], 0, 0, 0);
                }
            }
        } else if constexpr (X3) {
            // chunks 0..3 = hi of channels 8c..8c+7, chunks 4..7 = lo; two 16-channel k-steps per tap
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ah[m] = *reinterpret_cast<const uint4*>(bA + a_off[m] + ((ks << 5) ^ sw));
                    if constexpr (LO) al[m] = *reinterpret_cast<const uint4*>(bA + a_off[m] + ((64 + (ks << 5)) ^ sw));
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    bh[n] = *reinterpret_cast<const uint4*>(bB + b_off[n] + ks * 2 * BN * 16);
                    if constexpr (LO) bl[n] = *reinterpret_cast<const uint4*>(bB + b_off[n] + (4 + ks * 2) * BN * 16);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        if constexpr (LO) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, al[m]), __builtin_bit_cast(bf16x8, bh[n]), acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, ah[m]), __builtin_bit_cast(bf16x8, bl[n]), acc[m][n], 0, 0, 0);
                        }
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, ah[m]), __builtin_bit_cast(bf16x8, bh[n]), acc[m][n], 0, 0, 0);
                    }
            }
        } else {
        uint4 fa[MT], fb[NT];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int m = 0; m < MT; ++m) fa[m] = *reinterpret_cast<const uint4*>(bA + a_off[m] + ((ks << 5) ^ sw));
#pragma unroll
            for (int n = 0; n < NT; ++n) fb[n] = *reinterpret_cast<const uint4*>(bB + b_off[n] + ks * 2 * BN * 16);
            if constexpr (DT == MMR_DT_BF16) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, fa[m]), __builtin_bit_cast(bf16x8, fb[n]), acc[m][n], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                __builtin_bit_cast(f32x4, fa[m])[j], __builtin_bit_cast(f32x4, fb[n])[j], acc[m][n], 0, 0, 0);
            }
        }
        }
        MMR_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA of tap g+1 has landed (this wave's share)
        MMR_STAMP(2);
        __syncthreads();                                  // ... everyone's has; buffer `cur` is free
        MMR_STAMP(3);
#ifdef MMR_DIAG
        if constexpr (STAMP) st_acc[5] += 1;
#endif
        cur ^= 1;
        ++tap;
        if (tap == TAPS) {
            tap = 0;
            ++s;
            if (s < nslices && more) {  // every wave is past its last read of sA: install the next slice
                if constexpr (DMA_A) {
                    dma_stage_a(s);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    MMR_STAGE_A_REGS(s);
                }
                __syncthreads();
                MMR_STAMP(4);
            }
        }
    }
#ifdef MMR_DIAG
    if constexpr (STAMP) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) atomicAdd(&g_conv_stamp[wave][k], st_acc[k]);
        }
    }
#endif

    // ---- split-K: raw fp32 partial tile, finalised by conv_ksplit_finalize_kernel ----
    if (p.kpart) {
        if constexpr (M16) {
            if (p.kcompact) {
                float* kc = p.kpart + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * ((size_t)(TXT * TY * TZ) * p.Cout);
#pragma unroll
                for (int gi = 0; gi < NT / 2; ++gi) {
                    const int co = ntile * BN + wn * NT * 32 + gi * 64 + q16 * 16;
#pragma unroll
                    for (int mi = 0; mi < 2 * MT; ++mi) {
                        const int mt = wm * MT + (mi >> 1);
                        const int lx = mt >> 1, ly = (mt & 1) * 4 + 2 * (mi & 1) + (r16 >> 3), lz = r16 & 7;
                        if (co + 15 < p.Cout && x0 + lx < p.X && y0 + ly < p.Y && z0 + lz < p.Z) {
                            float* o = kc + (size_t)((lx * TY + ly) * TZ + lz) * p.Cout + co;
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni)
                                *reinterpret_cast<float4*>(o + ni * 4) = make_float4(acc16[mi][gi * 4 + ni][0], acc16[mi][gi * 4 + ni][1],
                                                                                   acc16[mi][gi * 4 + ni][2], acc16[mi][gi * 4 + ni][3]);
                        }
                    }
                }
                return;
            }
        }
        float* kp = p.kpart + (size_t)blockIdx.z * ((size_t)p.B * p.X * p.Y * p.Z * p.Cout);
        if constexpr (M16) {
#pragma unroll
            for (int gi = 0; gi < NT / 2; ++gi) {
                const int co = ntile * BN + wn * NT * 32 + gi * 64 + q16 * 16;
#pragma unroll
                for (int mi = 0; mi < 2 * MT; ++mi) {
                    const int mt = wm * MT + (mi >> 1);
                    const int gx = x0 + (mt >> 1), gy = y0 + (mt & 1) * 4 + 2 * (mi & 1) + (r16 >> 3), gz = z0 + (r16 & 7);
                    if (gx < p.X && gy < p.Y && gz < p.Z) {
                        const size_t o = ((((size_t)b * p.X + gx) * p.Y + gy) * p.Z + gz) * p.Cout + co;
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (co + ni * 4 + r < p.Cout) kp[o + ni * 4 + r] = acc16[mi][gi * 4 + ni][r];
                    }
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = ntile * BN + (wn * NT + n) * 32 + (lane & 31);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int mt = wm * MT + m;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int q = row_perm((r & 3) + 8 * (r >> 2) + 4 * h);
                        const int gx = x0 + (mt >> 1), gy = y0 + (mt & 1) * 4 + (q >> 3), gz = z0 + (q & 7);
                        if (co < p.Cout && gx < p.X && gy < p.Y && gz < p.Z)
                            kp[((((size_t)b * p.X + gx) * p.Y + gy) * p.Z + gz) * p.Cout + co] = acc[m][n][r];
                    }
                }
            }
        }
        return;
    }

    // ---- epilogue: bias + LeakyReLU, store ----
    const bool store_f32 = (DT != MMR_DT_BF16) || p.out_f32;
    float* s_col = reinterpret_cast<float*>(smem);  // [WM][BN] column sums (sA is free after the last barrier)
    if constexpr (M16) {
        // Physical weight column ni*16 + i of this wave's 64-column group holds cout (i >> 2)*16 + ni*4 + (i & 3)
        // (pack_kernel, conv_cout_of_col): lane (r16, q16) therefore owns the 16 CONSECUTIVE couts q16*16 .. +15
        // of voxel r16 across its four accumulator tiles -> 32-B (bf16) / 64-B (fp32) per lane, full 128-B lines.
        static_assert(NT == 2, "16x16x32 path: 64 columns per wave");
        const int cl = wn * 64 + q16 * 16;
        const int co = ntile * BN + cl;
        if constexpr (UPF) {
            // raw fp32 sums of parity class cls to voxel 2 i + p of the full-resolution partial tensor [B,2X,2Y,2Z,Cout]
            const int px = (cls >> 2) & 1, py = (cls >> 1) & 1, pz = cls & 1;
#pragma unroll
            for (int mi = 0; mi < 2 * MT; ++mi) {
                const int mt = wm * MT + (mi >> 1);
                const int lx = x0 + (mt >> 1), ly = y0 + (mt & 1) * 4 + 2 * (mi & 1) + (r16 >> 3), lz = z0 + (r16 & 7);
                if (co + 15 < p.Cout && lx < p.X && ly < p.Y && lz < p.Z) {
                    const size_t o = ((((size_t)b * (2 * p.X) + 2 * lx + px) * (2 * p.Y) + 2 * ly + py) * (2 * p.Z) + 2 * lz + pz) * p.Cout + co;
                    if constexpr (PART16) {
                        _Float16* po = reinterpret_cast<_Float16*>(p.out) + o;
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            f16x8 hv;
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                hv[e] = (_Float16)fminf(fmaxf(acc16[mi][2 * k + (e >> 2)][e & 3], -65504.f), 65504.f);
                            *reinterpret_cast<f16x8*>(po + 8 * k) = hv;
                        }
                    } else {
                        float* po = reinterpret_cast<float*>(p.out) + o;
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            *reinterpret_cast<float4*>(po + ni * 4) =
                                make_float4(acc16[mi][ni][0], acc16[mi][ni][1], acc16[mi][ni][2], acc16[mi][ni][3]);
                    }
                }
            }
            return;
        }
        const int cs = p.csplit;
        const bool second = cs > 0 && co >= cs;
        const int ostride = cs ? (second ? p.Cout - cs : cs) : p.Cout;  // row stride and column of this lane's block
        const int ocol = second ? co - cs : co;
        char* const optr = second ? p.out1 : p.out;
        const bool domask = p.ymask && (!cs || second);
        const bool vec = (co + 15 < p.Cout) && !(ostride & 7) && !(cs & 15);
        float bv[4][4], csum[4][4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (BIAS_PRE) bv[ni][r] = bias_pre[ni * 4 + r];
                else bv[ni][r] = (p.bias && co + ni * 4 + r < p.Cout) ? p.bias[co + ni * 4 + r] : 0.f;
                csum[ni][r] = 0.f;
            }
        // CV_POOLF: bit c of rt[mi] = "this lane's voxel of tile mi is the first maximum of its 2x2x2 window of ymask in channel
        // co + c" (window order x, y, z like MaxPooling3D's argmax / maxpool_bwd_v4_kernel).  Every lane takes part (DPP): lanes
        // whose voxel lies outside a ragged tile read a clamped address, their whole window is outside (even dims) and unused.
        unsigned rt[M16 ? 2 * MT : 1];
        if constexpr (POOLF) {
            float ymv[4][16];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int gx = x0 + t16x(mi), gy = y0 + t16y(mi) + (r16 >> 3), gz = z0 + (r16 & 7);
                const int cx = gx < p.X ? gx : p.X - 1, cy = gy < p.Y ? gy : p.Y - 1, cz = gz < p.Z ? gz : p.Z - 1;
                const float* ym = p.ymask + ((((size_t)b * p.X + cx) * p.Y + cy) * p.Z + cz) * p.Cout + co;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const float4 q = *reinterpret_cast<const float4*>(ym + ni * 4);
                    ymv[mi][ni * 4] = q.x; ymv[mi][ni * 4 + 1] = q.y; ymv[mi][ni * 4 + 2] = q.z; ymv[mi][ni * 4 + 3] = q.w;
                }
                rt[mi] = 0u;
            }
            const int ia = ((r16 >> 3) << 1) | (r16 & 1), ib = 4 + ia;     // window index of this lane's voxel in the x = 0 / x = 1 tile
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float a = ymv[pr][c], bq = ymv[pr + 2][c];
                    float m = fmaxf(a, bq);
                    m = fmaxf(m, dpp_xor1(m));
                    m = fmaxf(m, dpp_xor8(m));
                    int f = a == m ? ia : (bq == m ? ib : 8);
                    f = min(f, dpp_xor1(f));
                    f = min(f, dpp_xor8(f));
                    rt[pr] |= (f == ia ? 1u : 0u) << c;
                    rt[pr + 2] |= (f == ib ? 1u : 0u) << c;
                }
        }
#pragma unroll
        for (int mi = 0; mi < 2 * MT; ++mi) {
            const int gx = x0 + t16x(mi), gy = y0 + t16y(mi) + (r16 >> 3), gz = z0 + (r16 & 7);
            if (co < p.Cout && gx < p.X && gy < p.Y && gz < p.Z) {
                const size_t o = ((((size_t)b * p.X + gx) * p.Y + gy) * p.Z + gz) * ostride + ocol;
                float val[4][4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        val[ni][r] = acc16[mi][ni][r] + bv[ni][r];
                        if (p.leaky && val[ni][r] < 0.f) val[ni][r] *= p.alpha;
                    }
                if constexpr (POOLF) {   // + the pooling's gradient where this voxel was the window's maximum
                    const float* dp = p.dpool + ((((size_t)b * (p.X >> 1) + (gx >> 1)) * (p.Y >> 1) + (gy >> 1)) * (p.Z >> 1) + (gz >> 1)) * p.Cout + co;
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const float4 q = *reinterpret_cast<const float4*>(dp + ni * 4);
                        const unsigned bits = rt[mi] >> (ni * 4);
                        val[ni][0] += (bits & 1u) ? q.x : 0.f;
                        val[ni][1] += (bits & 2u) ? q.y : 0.f;
                        val[ni][2] += (bits & 4u) ? q.z : 0.f;
                        val[ni][3] += (bits & 8u) ? q.w : 0.f;
                    }
                }
                if (vec) {
                    if (domask) {
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) {
                            const float4 ym = *reinterpret_cast<const float4*>(p.ymask + o + ni * 4);
                            if (ym.x < 0.f) val[ni][0] *= p.alpha;
                            if (ym.y < 0.f) val[ni][1] *= p.alpha;
                            if (ym.z < 0.f) val[ni][2] *= p.alpha;
                            if (ym.w < 0.f) val[ni][3] *= p.alpha;
#pragma unroll
                            for (int r = 0; r < 4; ++r) csum[ni][r] += val[ni][r];
                        }
                    }
                    if (store_f32) {
                        float* po = reinterpret_cast<float*>(optr) + o;
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            *reinterpret_cast<float4*>(po + ni * 4) = make_float4(val[ni][0], val[ni][1], val[ni][2], val[ni][3]);
                    } else {
                        bf16_t* po = reinterpret_cast<bf16_t*>(optr) + o;
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            uint4 pk;
                            pk.x = (unsigned)f32_to_bf16(val[2 * k][0]) | ((unsigned)f32_to_bf16(val[2 * k][1]) << 16);
                            pk.y = (unsigned)f32_to_bf16(val[2 * k][2]) | ((unsigned)f32_to_bf16(val[2 * k][3]) << 16);
                            pk.z = (unsigned)f32_to_bf16(val[2 * k + 1][0]) | ((unsigned)f32_to_bf16(val[2 * k + 1][1]) << 16);
                            pk.w = (unsigned)f32_to_bf16(val[2 * k + 1][2]) | ((unsigned)f32_to_bf16(val[2 * k + 1][3]) << 16);
                            *reinterpret_cast<uint4*>(po + 8 * k) = pk;
                        }
                    }
                } else {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int e = ni * 4 + r;
                            if (co + e < p.Cout) {
                                if (domask) {
                                    if (p.ymask[o + e] < 0.f) val[ni][r] *= p.alpha;
                                    csum[ni][r] += val[ni][r];
                                }
                                if (store_f32) reinterpret_cast<float*>(optr)[o + e] = val[ni][r];
                                else reinterpret_cast<bf16_t*>(optr)[o + e] = f32_to_bf16(val[ni][r]);
                            }
                        }
                }
            }
        }
        if (p.ymask) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float c = csum[ni][r];
                    c += __shfl_xor(c, 1);
                    c += __shfl_xor(c, 2);
                    c += __shfl_xor(c, 4);
                    c += __shfl_xor(c, 8);
                    if (r16 == 0) s_col[wm * BN + cl + ni * 4 + r] = c;
                }
        }
        if (p.ymask) {
            __syncthreads();
            const int col = ntile * BN + tid;
            if (tid < BN && col < p.Cout && col >= p.csplit) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < WM; ++k) t += (double)s_col[k * BN + tid];
                p.part[(size_t)(col - p.csplit) * gridDim.x + blockIdx.x] = t;   // [column][tile]: colsum_final_kernel reads rows
            }
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = ntile * BN + (wn * NT + n) * 32 + (lane & 31);
        const bool cok = co < p.Cout;
        const float bv = (cok && p.bias) ? p.bias[co] : 0.f;
        float csum = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int mt = wm * MT + m;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = row_perm((r & 3) + 8 * (r >> 2) + 4 * h);
                const int gx = x0 + (mt >> 1), gy = y0 + (mt & 1) * 4 + (q >> 3), gz = z0 + (q & 7);
                if (cok && gx < p.X && gy < p.Y && gz < p.Z) {
                    float val = acc[m][n][r] + bv;
                    if (p.leaky && val < 0.f) val *= p.alpha;
                    const size_t o = ((((size_t)b * p.X + gx) * p.Y + gy) * p.Z + gz) * p.Cout + co;
                    if (p.ymask) {
                        if (p.ymask[o] < 0.f) val *= p.alpha;
                        csum += val;
                    }
                    if (store_f32) reinterpret_cast<float*>(p.out)[o] = val;
                    else reinterpret_cast<bf16_t*>(p.out)[o] = f32_to_bf16(val);
                }
            }
        }
        if (p.ymask) {
            csum += __shfl_xor(csum, 32);
            if (h == 0) s_col[wm * BN + (wn * NT + n) * 32 + (lane & 31)] = csum;
        }
    }
    if (p.ymask) {
        __syncthreads();
        if (tid < BN && ntile * BN + tid < p.Cout) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < WM; ++k) t += (double)s_col[k * BN + tid];
            p.part[(size_t)(ntile * BN + tid) * gridDim.x + blockIdx.x] = t;
        }
    }
}

// one wave per channel over part[channel][tile] (the conv epilogues store their column sums transposed, so that a channel's
// partials are one contiguous row: whole 512-B lines per wave load; with [tile][channel] every load touched 64 lines and the
// 4 MB of a full-resolution launch took 27 us per call, 6 calls per training step): strided partial sums in a fixed order, then
// a wave reduction -> reproducible
__global__ void __launch_bounds__(64)
colsum_final_kernel(const double* __restrict__ part, float* __restrict__ db, int C, int nblk, int accumulate)
{
    const int c = blockIdx.x;
    const double* row = part + (size_t)c * nblk;
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;      // four loads in flight per lane, combined in a fixed order
    int k = threadIdx.x;
    for (; k + 192 < nblk; k += 256) {
        r0 += row[k]; r1 += row[k + 64]; r2 += row[k + 128]; r3 += row[k + 192];
    }
    for (; k < nblk; k += 64) r0 += row[k];
    double r = (r0 + r1) + (r2 + r3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o);
    if (threadIdx.x == 0) {
        if (accumulate) db[c] += (float)r; else db[c] = (float)r;
    }
}

// ---- weight packing ------------------------------------------------------ //
__host__ __device__ inline int conv_bn(int Cout)
{
    if (Cout % 256 == 0) return 256;
    if (Cout % 128 == 0) return 128;
    if (Cout % 64 == 0) return 64;
    return 32;
}

// The 16x16x32 kernels (every dtype but exact fp32, N tiles of 64 columns and wider) keep their weight columns
// permuted inside each wave's 64-column group so that the transposed accumulator layout hands every lane 16
// consecutive couts (see the epilogue): physical column ni*16 + i  <->  cout (i >> 2)*16 + ni*4 + (i & 3).
__host__ __device__ inline bool conv_uses_m16(int dtype, int BN) { return dtype != MMR_DT_F32 && BN >= 64; }
__host__ __device__ inline int conv_cout_of_col(int col)
{
    const int g = col >> 6, ni = (col >> 4) & 3, i = col & 15;
    return (g << 6) + ((i >> 2) << 4) + (ni << 2) + (i & 3);
}

// One 16-B chunk (item i) of a weight image.  ``ld`` / ``off``: the job reads input-channel rows off .. off + n of a Keras
// kernel whose row count is ld ([27][ld][cols]), so a channel slice of a concat layer needs no contiguous copy.
template <int DT>
__device__ __forceinline__ void pack_item(const float* __restrict__ w, char* __restrict__ wp, int Cin, int Cout, int BN,
                                          int nslices, int transpose_flip, int plain_cols, int ld, int off, int64_t i)
{
    constexpr int KC = Elt<DT>::kc;
    constexpr int EPC = (DT == MMR_DT_F32) ? 4 : 8;  // elements per 16-B chunk (fp32x3 chunks hold 8 bf16)
    int64_t r = i;
    const int col = (int)(r % BN); r /= BN;
    const int chunk = (int)(r % 8); r /= 8;
    const int tap = (int)(r % 27); r /= 27;
    const int s = (int)(r % nslices);
    const int t = (int)(r / nslices);
    const int co = t * BN + ((conv_uses_m16(DT, BN) && !plain_cols) ? conv_cout_of_col(col) : col);
    char* dst = wp + i * 16;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int cc = (DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1) ? (chunk & 3) : chunk;
        const int ci = s * KC + cc * EPC + e;
        float v = 0.f;
        if (co < Cout) {
            if (transpose_flip) v = w[((int64_t)(26 - tap) * ld + off + co) * Cin + ci];  // keras dims [27][ld >= Cout][Cin]
            else v = w[((int64_t)tap * ld + off + ci) * Cout + co];
        }
        if (DT == MMR_DT_F32) {
            reinterpret_cast<float*>(dst)[e] = v;
        } else {
            bf16_t hb = f32_to_bf16(v);
            if ((DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1) && chunk >= 4) hb = f32_to_bf16(v - bf16_to_f32(hb));  // lo part
            reinterpret_cast<bf16_t*>(dst)[e] = hb;
        }
    }
}

template <int DT>
__global__ void pack_kernel(const float* __restrict__ w, char* __restrict__ wp, int Cin, int Cout, int BN, int ntiles,
                            int transpose_flip, int plain_cols)
{
    const int nslices = Cin / Elt<DT>::kc;
    const int64_t total = (int64_t)ntiles * nslices * 27 * 8 * BN;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        pack_item<DT>(w, wp, Cin, Cout, BN, nslices, transpose_flip, plain_cols, transpose_flip ? Cout : Cin, 0, i);
}

// Folded-upsampling weight image (CV_UPFOLD): [class 8][n-tile][slice][tap 8][chunk 8][BN][16 B]; tap bit s per axis of
// parity class bit p sums the original taps { p=0,s=0: {0} | p=0,s=1: {1,2} | p=1,s=0: {0,1} | p=1,s=1: {2} } (fp32 sums,
// then the same bf16 / hi-lo encoding and column permutation as pack_kernel).  w = Keras [27][C0][Cout] of the upsampled
// channels only.
template <int DT>
__device__ __forceinline__ void pack_upfold_item(const float* __restrict__ w, char* __restrict__ wp, int Cin, int Cout, int BN,
                                                 int ntiles, int nslices, int ld, int off, int64_t i)
{
    constexpr int KC = Elt<DT>::kc;
    constexpr int EPC = 8;
    int64_t r = i;
    const int col = (int)(r % BN); r /= BN;
    const int chunk = (int)(r % 8); r /= 8;
    const int tap = (int)(r % 8); r /= 8;
    const int s = (int)(r % nslices); r /= nslices;
    const int t = (int)(r % ntiles);
    const int cls = (int)(r / ntiles);
    const int co = t * BN + conv_cout_of_col(col);
    int lo3[3], n3[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int pb = (cls >> (2 - a)) & 1, sb = (tap >> (2 - a)) & 1;
        lo3[a] = pb == 0 ? (sb == 0 ? 0 : 1) : (sb == 0 ? 0 : 2);
        n3[a] = (pb == 0) == (sb == 0) ? 1 : 2;        // (0,0) and (1,1): one tap; (0,1) and (1,0): two
    }
    char* dst = wp + i * 16;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int cc = (DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1) ? (chunk & 3) : chunk;
        const int ci = s * KC + cc * EPC + e;
        float v = 0.f;
        if (co < Cout) {
            for (int a = 0; a < n3[0]; ++a)
                for (int bb = 0; bb < n3[1]; ++bb)
                    for (int c = 0; c < n3[2]; ++c) {
                        const int tp = ((lo3[0] + a) * 3 + (lo3[1] + bb)) * 3 + (lo3[2] + c);
                        v += w[((int64_t)tp * ld + off + ci) * Cout + co];
                    }
        }
        bf16_t hb = f32_to_bf16(v);
        if ((DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1) && chunk >= 4) hb = f32_to_bf16(v - bf16_to_f32(hb));  // lo part
        reinterpret_cast<bf16_t*>(dst)[e] = hb;
    }
}

template <int DT>
__global__ void pack_upfold_kernel(const float* __restrict__ w, char* __restrict__ wp, int Cin, int Cout, int BN, int ntiles)
{
    const int nslices = Cin / Elt<DT>::kc;
    const int64_t total = (int64_t)8 * ntiles * nslices * 8 * 8 * BN;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        pack_upfold_item<DT>(w, wp, Cin, Cout, BN, ntiles, nslices, Cin, 0, i);
}

// Weight image of the dgrad fold (CV_DGFOLD): [n-tile over the C0 gradient channels][class 8][dz-channel slice][tap 8][chunk]
// [BN][16 B], value(k = dz channel co, n = up channel c) = sum of the forward weights W[t][c][co] over the taps t that
// (class, tap bit) covers (same table as pack_upfold_kernel; no tap flip: the halo offsets 2 - p - s carry the transpose).
// w = Keras [27][C0][Cz] of the upsampled input channels (Cz = the layer's output channels = dz channels).
template <int DT>
__device__ __forceinline__ void pack_dgfold_item(const float* __restrict__ w, char* __restrict__ wp, int C0, int Cz, int BN,
                                                 int ncs, int ld, int off, int64_t i)
{
    constexpr int KC = Elt<DT>::kc;
    int64_t r = i;
    const int col = (int)(r % BN); r /= BN;
    const int chunk = (int)(r % 8); r /= 8;
    const int tap = (int)(r % 8); r /= 8;
    const int cs = (int)(r % ncs); r /= ncs;
    const int cls = (int)(r % 8);
    const int t = (int)(r / 8);
    const int n = t * BN + (conv_uses_m16(DT, BN) ? conv_cout_of_col(col) : col);   // up channel c
    int lo3[3], n3[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int pb = (cls >> (2 - a)) & 1, sb = (tap >> (2 - a)) & 1;
        lo3[a] = pb == 0 ? (sb == 0 ? 0 : 1) : (sb == 0 ? 0 : 2);
        n3[a] = (pb == 0) == (sb == 0) ? 1 : 2;
    }
    char* dst = wp + i * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = cs * KC + (chunk & 3) * 8 + e;                                 // dz channel co
        float v = 0.f;
        if (n < C0) {
            for (int a = 0; a < n3[0]; ++a)
                for (int bb = 0; bb < n3[1]; ++bb)
                    for (int c = 0; c < n3[2]; ++c) {
                        const int tp = ((lo3[0] + a) * 3 + (lo3[1] + bb)) * 3 + (lo3[2] + c);
                        v += w[((int64_t)tp * ld + off + n) * Cz + k];
                    }
        }
        bf16_t hb = f32_to_bf16(v);
        if (chunk >= 4) hb = f32_to_bf16(v - bf16_to_f32(hb));  // lo part
        reinterpret_cast<bf16_t*>(dst)[e] = hb;
    }
}

template <int DT>
__global__ void pack_dgfold_kernel(const float* __restrict__ w, char* __restrict__ wp, int C0, int Cz, int BN, int ntiles)
{
    const int ncs = Cz / Elt<DT>::kc;
    const int64_t total = (int64_t)ntiles * 8 * ncs * 8 * 8 * BN;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        pack_dgfold_item<DT>(w, wp, C0, Cz, BN, ncs, C0, 0, i);
}

// All weight images of a training step in ONE launch (mmr_conv3d_k3_pack_batch): item i of the launch belongs to the job
// whose [first, first + items) holds it; the job table travels as a kernel argument.
struct PackJobDev { const float* w; char* out; int64_t first; int kind, ld, off, a, b, BN, ntiles, nsl; };
constexpr int PACK_BATCH_MAX = 32;
struct PackBatch { int n; int64_t total; PackJobDev j[PACK_BATCH_MAX]; };

template <int DT>
__global__ void __launch_bounds__(256) pack_batch_kernel(const PackBatch pb)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pb.total; i += (int64_t)gridDim.x * blockDim.x) {
        int jj = 0;
        while (jj + 1 < pb.n && i >= pb.j[jj + 1].first) ++jj;
        const PackJobDev& q = pb.j[jj];
        const int64_t li = i - q.first;
        if (q.kind == MMR_PACK_FWD) pack_item<DT>(q.w, q.out, q.a, q.b, q.BN, q.nsl, 0, 0, q.ld, q.off, li);
        else if (q.kind == MMR_PACK_DGRAD) pack_item<DT>(q.w, q.out, q.b, q.a, q.BN, q.nsl, 1, 0, q.ld, q.off, li);
        else if constexpr (DT != MMR_DT_F32) {
            if (q.kind == MMR_PACK_UPFOLD) pack_upfold_item<DT>(q.w, q.out, q.a, q.b, q.BN, q.ntiles, q.nsl, q.ld, q.off, li);
            else pack_dgfold_item<DT>(q.w, q.out, q.a, q.b, q.BN, q.nsl, q.ld, q.off, li);
        }
    }
}

// out = act(sum_k kpart[k] + bias), summed in index order (bitwise reproducible), fp32 or bf16 store
__global__ void __launch_bounds__(256)
conv_ksplit_finalize_kernel(const float* __restrict__ kpart, int nk, const float* __restrict__ bias, void* __restrict__ out,
                            int64_t n, int Cout, int leaky, float alpha, int out_bf16)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int k = 0; k < nk; ++k) v += kpart[(int64_t)k * n + i];
        if (bias) v += bias[i % Cout];
        if (leaky && v < 0.f) v *= alpha;
        if (out_bf16) reinterpret_cast<bf16_t*>(out)[i] = f32_to_bf16(v);
        else reinterpret_cast<float*>(out)[i] = v;
    }
}

// Tail of a launch whose tile count is not a multiple of the CU count (one workgroup per CU: a launch of 2 400 tiles runs as
// ten rounds of 256, the last one 37 % full -- 9.4 rounds of work in the time of 10; 300 tiles: 1.2 in the time of 2).  The
// last nblk % ncu tiles are launched on their own with their K walk split over ncu / R workgroups each (compact fp32 partials,
// fixed-order sum -> bitwise reproducible), so that the partial round takes 1 / St of a tile time.
inline int conv_ncu()
{
    static int ncu = 0;
    if (!ncu) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
            n = 256;
        ncu = n;
    }
    return ncu;
}
inline bool conv_tail_plan(int64_t nblk, int ntiles_n, int G, int* R_out, int* S_out)
{
    const int ncu = conv_ncu();
    // worth it from four 27-tap slices per tile on (a 64 -> 64 fp32x3 tile, G = 54, is so short that the second launch, the
    // partials and the per-block A staging cost what the shorter round saves: 2.04 -> 2.07 ms at 160^3), and every block walks
    // at least one whole slice
    if (ntiles_n != 1 || nblk <= ncu || G < 108) return false;
    const int R = (int)(nblk % ncu);
    if (R == 0 || 2 * R > ncu) return false;
    int S = ncu / R;
    if (S > G / 27) S = G / 27;
    if (S > 16) S = 16;
    if (S < 2) return false;
    *R_out = R;
    *S_out = S;
    return true;
}

// out (tiles tile0 ..) = act(sum_k kpart[k][tile][voxel][:] + bias), summed in index order; TV = voxels per tile (TXT x 8 x 8)
__global__ void __launch_bounds__(256)
conv_ktail_finalize_kernel(const float* __restrict__ kpart, int nk, int R, int tile0, const float* __restrict__ bias,
                           void* __restrict__ out, int X, int Y, int Z, int Cout, int ntx, int nty, int ntz, int TXT_,
                           int leaky, float alpha, int out_bf16)
{
    const int TV = TXT_ * TY * TZ, c4n = Cout / 4;
    const int64_t total = (int64_t)R * TV * c4n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        const int vloc = (int)((i / c4n) % TV);
        const int t = (int)(i / ((int64_t)c4n * TV));
        int bid = tile0 + t;
        const int tzi = bid % ntz; bid /= ntz;
        const int tyi = bid % nty; bid /= nty;
        const int txi = bid % ntx;
        const int b = bid / ntx;
        const int gx = txi * TXT_ + vloc / (TY * TZ), gy = tyi * TY + (vloc / TZ) % TY, gz = tzi * TZ + vloc % TZ;
        if (gx >= X || gy >= Y || gz >= Z) continue;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < nk; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(kpart + (((size_t)k * R + t) * TV + vloc) * Cout + c4 * 4);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if (bias) {
            const float4 bq = *reinterpret_cast<const float4*>(bias + c4 * 4);
            v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
        }
        if (leaky) {
            v.x = v.x < 0.f ? v.x * alpha : v.x; v.y = v.y < 0.f ? v.y * alpha : v.y;
            v.z = v.z < 0.f ? v.z * alpha : v.z; v.w = v.w < 0.f ? v.w * alpha : v.w;
        }
        const size_t o = ((((size_t)b * X + gx) * Y + gy) * Z + gz) * Cout + c4 * 4;
        if (out_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + o) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        else *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + o) = v;
    }
}

// Split the K walk when the launch has fewer workgroups than half the CUs and a scratch buffer is offered.
inline int conv_ksplit(int64_t nblk, int ntiles_n, int G)
{
    const int64_t wgs = nblk * ntiles_n;
    if (wgs >= 128 || G < 18) return 1;
    int smax = G / 9;          // at least 9 taps per block: the A tile is staged per block
    if (smax > 16) smax = 16;
    // rounds of workgroups (one per CU) x (taps per block + what a block costs before its first tap: A tile, offset table,
    // first weights ~ 12 taps).  The round count matters: 38 tiles x 7 = 266 blocks is TWO rounds of a seventh each, x 6 is one
    // round of a sixth (the old rule, ceil(256 / tiles), overshot the chip by a few blocks at 38, 43, 52, 125 ... tiles).
    const int ncu = conv_ncu();
    int best = 1;
    int64_t best_cost = 0;
    for (int S = 1; S <= (smax < 1 ? 1 : smax); ++S) {
        const int64_t rounds = (wgs * S + ncu - 1) / ncu;
        const int64_t cost = rounds * ((G + S - 1) / S + 12);
        if (S == 1 || cost < best_cost) { best = S; best_cost = cost; }
    }
    return best;
}

template <int DT, int WM, int WN, int MT, int NT, int VAR>
int launch_conv(const ConvParams& p, int ntiles_n, hipStream_t st, int64_t* nblk_out)
{
    constexpr int BN = WN * NT * 32;
    constexpr int TXT = WM * MT * 32 / (TY * TZ);
    // + the item -> offset table of the bf16 256-column tile (ATAB in the kernel): 600 rows x 8 chunks x 4 B
    constexpr bool ATAB_L = DT == MMR_DT_BF16 && MT == 4 && TXT == 4 &&
                            (VAR & (CV_M16 | CV_PIPE | CV_DMA_A)) == (CV_M16 | CV_PIPE | CV_DMA_A) && (VAR & CV_DGFOLD) == 0;
    constexpr int LDS = (TXT + 2) * HY * HZ * ROWB + 2 * BN * 128 + (ATAB_L ? (TXT + 2) * HY * HZ * 8 * 4 : 0);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    auto kern = conv3d_k3_kernel<DT, WM, WN, MT, NT, VAR>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
        attr_set = true;
    }
    ConvParams q = p;
    q.ntx = (p.X + TXT - 1) / TXT;
    const int64_t nblk = (int64_t)q.B * q.ntx * q.nty * q.ntz;
    if (nblk > 0x7fffffff) return MMR_EINVAL;
    if (nblk_out) *nblk_out = nblk;
    const int G = ((p.C0 + p.C1) / Elt<DT>::kc) * 27;
    const int S = (p.kpart && !p.ymask) ? conv_ksplit(nblk, ntiles_n, G) : 1;
    if (S > 1) {
        q.gsplit = (G + S - 1) / S;
        const int nz = (G + q.gsplit - 1) / q.gsplit;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk, ntiles_n, nz), dim3(CONV_THREADS), LDS, st, q);
        const int64_t n = (int64_t)p.B * p.X * p.Y * p.Z * p.Cout;
        const bool obf = (DT == MMR_DT_BF16) && !p.out_f32;
        hipLaunchKernelGGL(conv_ksplit_finalize_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, st, (const float*)p.kpart, nz,
                           p.bias, (void*)p.out, n, p.Cout, p.leaky, p.alpha, obf ? 1 : 0);
        return check_launch();
    }
    // tail split (see conv_tail_plan): 16x16x32 kernels with the plain epilogue, a workspace offered by the caller
    constexpr bool TAIL_OK = ((VAR & CV_M16) != 0) && DT != MMR_DT_F32 && NT == 2 && (VAR & (CV_UPFOLD | CV_DGFOLD)) == 0;
    int R = 0, St = 0;
    if (TAIL_OK && p.kpart && !p.ymask && !p.csplit && (p.Cout & 15) == 0 && conv_tail_plan(nblk, ntiles_n, G, &R, &St)) {
        ConvParams m = q;
        m.kpart = nullptr;
        hipLaunchKernelGGL(kern, dim3((unsigned)(nblk - R), ntiles_n), dim3(CONV_THREADS), LDS, st, m);
        ConvParams t = q;
        t.tile0 = (int)(nblk - R);
        t.kcompact = 1;
        t.gsplit = (G + St - 1) / St;
        const int nz = (G + t.gsplit - 1) / t.gsplit;
        hipLaunchKernelGGL(kern, dim3((unsigned)R, ntiles_n, nz), dim3(CONV_THREADS), LDS, st, t);
        const bool obf = (DT == MMR_DT_BF16) && !p.out_f32;
        const int64_t n4 = (int64_t)R * (TXT * TY * TZ) * (p.Cout / 4);
        hipLaunchKernelGGL(conv_ktail_finalize_kernel, dim3(stream_grid(n4, 256)), dim3(256), 0, st, (const float*)p.kpart, nz, R,
                           t.tile0, p.bias, (void*)p.out, p.X, p.Y, p.Z, p.Cout, q.ntx, q.nty, q.ntz, TXT, p.leaky, p.alpha,
                           obf ? 1 : 0);
        return check_launch();
    }
    q.kpart = nullptr;
    // more than two rounds of workgroups: below that the classes of a tile run side by side anyway
    if ((VAR & CV_UPFOLD) != 0) q.xcd_pair = (nblk * ntiles_n * 8 > 2 * (int64_t)conv_ncu()) ? 4 : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, ((VAR & CV_UPFOLD) != 0 ? 8 : 1) * ntiles_n), dim3(CONV_THREADS), LDS, st, q);
    return check_launch();
}

template <int DT>
int dispatch_conv(const ConvParams& p, hipStream_t st, int64_t* nblk_out = nullptr)
{
    const int BN = conv_bn(p.Cout);
    const int nt = (p.Cout + BN - 1) / BN;
    // Per-tile defaults, each measured against its alternatives on one box, back to back:
    //  BN 256 bf16: explicit fragment pipeline + A tile by LDS-DMA + weight DMA from inside waves 0-3's MFMA stream
    //         (ms per C2 pair: all waves at the top of the tap 44.2 | waves 4-7 at the top, 0-3 after their MFMAs 43.7 |
    //         waves 0-3 issue everything after their MFMAs 43.2 | this 42.1-42.7);
    //  BN 256 fp32 tensors: + batched branch-free A staging (-1.5 % on the 256 -> 256 layer of C2);
    //  BN 128: 8x8x8-voxel tiles; bf16: compiler-scheduled fragments (neither the pipeline nor the DMA staging fit 256
    //         VGPRs at this tile); others: pipeline + DMA / batched staging (-2.3 % on the C3 dgrad convs);
    //  BN 64: + static priority 1 for waves 4-7 (-1 % on the 64 -> 64 layer at 160^3).
    constexpr bool F32T = (DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1);
    constexpr int V_FULL = CV_M16 | CV_PIPE | CV_DMA_A;
#ifdef MMR_DIAG
    const bool stamps = g_diag_stamps != 0;   // mmr_debug_set_stamps(1), tools/conv_stamps.py; not in the default build
    if (stamps && BN == 256) return launch_conv<DT, 2, 4, 4, 2, V_FULL | (F32T ? CV_BATCHA : 0) | CV_STAMP>(p, nt, st, nblk_out);
    if (stamps && BN == 64) return launch_conv<DT, 8, 1, 2, 2, V_FULL | CV_BATCHA | CV_PRIO_Y | CV_STAMP>(p, nt, st, nblk_out);
#endif
    if (p.dpool) {   // masked data gradient + MaxPooling3D backward in the epilogue (CV_POOLF): the 64-column fp32 tiles only
        if constexpr (F32T) {
            if (BN == 64 && p.ymask && !p.csplit && !((p.X | p.Y | p.Z) & 1))
                return launch_conv<DT, 8, 1, 2, 2, V_FULL | CV_BATCHA | CV_PRIO_Y | CV_POOLF>(p, nt, st, nblk_out);
        }
        return MMR_EUNSUPPORTED;
    }
    switch (BN) {
        case 256: return launch_conv<DT, 2, 4, 4, 2, V_FULL | (F32T ? CV_BATCHA : 0)>(p, nt, st, nblk_out);
        case 128:
            if (DT == MMR_DT_BF16) return launch_conv<DT, 4, 2, 4, 2, CV_M16>(p, nt, st, nblk_out);
            return launch_conv<DT, 4, 2, 4, 2, V_FULL | CV_BATCHA>(p, nt, st, nblk_out);
        case 64: return launch_conv<DT, 8, 1, 2, 2, V_FULL | CV_BATCHA | CV_PRIO_Y>(p, nt, st, nblk_out);
        default: return launch_conv<DT, 8, 1, 2, 1, CV_DMA_A>(p, nt, st, nblk_out);
    }
}

// The two halves of a folded-upsampling layer (EXTRA = CV_UPFOLD or CV_CINIT) on the per-tile default of the plain conv
template <int DT, int EXTRA>
int dispatch_conv_fold(const ConvParams& p, hipStream_t st, int64_t* nblk_out = nullptr)
{
    static_assert(DT == MMR_DT_BF16 || DT == MMR_DT_F32X3 || (DT == MMR_DT_F32X1 && (EXTRA & CV_DGFOLD) != 0),
                  "folded upsampling: bf16 / fp32x3 (dgrad also x1)");
    const int BN = conv_bn(p.Cout);
    const int nt = (p.Cout + BN - 1) / BN;
    constexpr bool F32T = (DT == MMR_DT_F32X3 || DT == MMR_DT_F32X1);
    static_assert(!((EXTRA & CV_PART16) != 0 && F32T), "half partial: bf16 layers only");
    constexpr int V_FULL = CV_M16 | CV_PIPE | CV_DMA_A;
    switch (BN) {
        case 256: return launch_conv<DT, 2, 4, 4, 2, V_FULL | (F32T ? CV_BATCHA : 0) | EXTRA>(p, nt, st, nblk_out);
        case 128:
            if constexpr (DT == MMR_DT_BF16) return launch_conv<DT, 4, 2, 4, 2, CV_M16 | EXTRA>(p, nt, st, nblk_out);
            else return launch_conv<DT, 4, 2, 4, 2, V_FULL | CV_BATCHA | EXTRA>(p, nt, st, nblk_out);
        case 64: return launch_conv<DT, 8, 1, 2, 2, V_FULL | (F32T ? CV_BATCHA : 0) | CV_PRIO_Y | EXTRA>(p, nt, st, nblk_out);
        default: return MMR_EUNSUPPORTED;
    }
}

#ifdef MMR_DIAG
// Diagnostic (-DMMR_DIAG builds only, not in mmr.h): switch the stamped instantiations on / off; copy out and clear
// the cycle stamps (tools/conv_stamps.py).
extern "C" int mmr_debug_set_stamps(int on) { g_diag_stamps = on; return MMR_OK; }
extern "C" int mmr_debug_conv_stamps(unsigned long long* out64)
{
    unsigned long long z[64] = {0};
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_conv_stamp), sizeof(z)) != hipSuccess) return MMR_EHIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamp), z, sizeof(z)) != hipSuccess) return MMR_EHIP;
    return MMR_OK;
}
#endif

// ---- first layer: concat(moving, fixed) (2 ch) -> Cout, VALU ------------- //
// Block = 4x4x16 voxel tile; thread = (cout, voxel-group); the thread's 54
// weights live in registers, the haloed 2-channel patch in LDS (broadcast reads).
constexpr int F_TX = 4, F_TY = 4, F_TZ = 16;
template <int OUT_DT>
__global__ void __launch_bounds__(256)
conv3d_cin2_kernel(const float* __restrict__ src, const float* __restrict__ trg, const float* __restrict__ w,
                   const float* __restrict__ bias, void* __restrict__ out, int B, int X, int Y, int Z, int Cout,
                   int leaky, float alpha, int ntx, int nty, int ntz)
{
    constexpr int PX = F_TX + 2, PY = F_TY + 2, PZ = F_TZ + 2;
    __shared__ float patch[PX][PY][PZ][2];
    int bid = blockIdx.x;
    const int tzi = bid % ntz; bid /= ntz;
    const int tyi = bid % nty; bid /= nty;
    const int txi = bid % ntx;
    const int b = bid / ntx;
    const int x0 = txi * F_TX, y0 = tyi * F_TY, z0 = tzi * F_TZ;
    const size_t nvox = (size_t)X * Y * Z;
    for (int i = threadIdx.x; i < PX * PY * PZ; i += 256) {
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
    const int ngroups = 256 / (Cout < 256 ? Cout : 256);  // voxel groups per block (Cout<=256, divides 256)
    for (int cobase = 0; cobase < Cout; cobase += 256) {
        const int co = cobase + (threadIdx.x % (Cout < 256 ? Cout : 256));
        const int grp = threadIdx.x / (Cout < 256 ? Cout : 256);
        float wr[54];
#pragma unroll
        for (int k = 0; k < 54; ++k) wr[k] = w[k * Cout + co];  // keras [27][2][Cout]
        const float bv = bias ? bias[co] : 0.f;
        for (int v = grp; v < F_TX * F_TY * F_TZ; v += ngroups) {
            const int vz = v % F_TZ, vy = (v / F_TZ) % F_TY, vx = v / (F_TZ * F_TY);
            float acc = bv;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dz = 0; dz < 3; ++dz) {
                        const float2 pv = *reinterpret_cast<const float2*>(&patch[vx + dx][vy + dy][vz + dz][0]);
                        const int t = (dx * 3 + dy) * 3 + dz;
                        acc = fmaf(pv.x, wr[t * 2], acc);
                        acc = fmaf(pv.y, wr[t * 2 + 1], acc);
                    }
            const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
            if (gx < X && gy < Y && gz < Z) {
                if (leaky && acc < 0.f) acc *= alpha;
                const size_t o = ((size_t)b * nvox + ((size_t)gx * Y + gy) * Z + gz) * Cout + co;
                if (OUT_DT == MMR_DT_BF16) reinterpret_cast<bf16_t*>(out)[o] = f32_to_bf16(acc);
                else reinterpret_cast<float*>(out)[o] = acc;
            }
        }
    }
}

// ---- first layer on the matrix cores --------------------------------------------------------------- //
// concat(moving, fixed) has 2 channels -> K = 27 taps x 2 = 54 (padded to 64).  Per 4x8x8 voxel tile the
// im2col matrix X[256 voxels][64 k] is built in LDS from a haloed image patch, the weights are an LDS image
// W[Cout][64 k], and D[cout][voxel] = W . X^T runs on v_mfma_f32_32x32x16_bf16 (bf16 mode: 4 MFMAs per
// 32x32 tile; fp32x3 mode: hi/lo splits, 12 MFMAs, ~5e-6 relative error).  With couts on the accumulator
// rows every lane owns runs of 4 consecutive couts of one voxel, so the epilogue stores 16 B per lane
// (bf16: after a v_permlane32_swap pairing of the two half-waves) -- the layer is bound by its
// N*Cout output write, not by the 136 GFLOP of arithmetic.
constexpr int M2_THREADS = 256;

// waves_per_eu(2, 3): with the default budget of a 256-thread block (512 registers per lane) hipcc puts the MFMA results
// in AGPRs and the epilogue pays a v_accvgpr_read per value; held to the occupancy the LDS allows anyway, the accumulators
// live in VGPRs (no AGPRs, no scratch)
template <bool X3, bool OUT_BF16>
__global__ void __launch_bounds__(M2_THREADS) __attribute__((amdgpu_waves_per_eu(2, 3)))
conv3d_cin2_mfma_kernel(const float* __restrict__ src, const float* __restrict__ trg, const float* __restrict__ w,
                        const float* __restrict__ bias, void* __restrict__ out, void* __restrict__ pool, int B, int X, int Y,
                        int Z, int Cout, int leaky, float alpha, int ntx, int nty, int ntz, int ntiles)
{
    // Voxel order of the im2col rows: row v = wave*64 + vt*32 + l  <->  x = 2*(wave >> 1) + vt, y = 4*(wave & 1) + (l >> 3),
    // z = l & 7.  A wave then owns both x planes of a 2x2x2 pooling window (its two accumulator tiles), the y pair is
    // lanes l / l ^ 8 and the z pair lanes l / l ^ 1, so the fused MaxPooling3D(2) is two DPP moves per value and no LDS.
    auto vox_x = [](int v) { return 2 * (v >> 7) + ((v >> 5) & 1); };
    auto vox_y = [](int v) { return 4 * ((v >> 6) & 1) + ((v >> 3) & 3); };
    auto vox_z = [](int v) { return v & 7; };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPL = X3 ? 2 : 1;
    float* s_img = reinterpret_cast<float*>(smem);                 // [600][2]
    float* s_bias = s_img + HROWS * 2;                             // [Cout] (<= 512)
    char* s_w = smem + HROWS * 8 + 2048;                           // [NPL][Cout][128 B]
    constexpr int ES_O = OUT_BF16 ? 2 : 4;
    constexpr int G = OUT_BF16 ? 2 : 1;                            // 32-cout groups per 128-B line of a voxel (Cout % (32 G) == 0)
    constexpr int ST_PITCH = 144, ST_ROWS = 72;                    // 64 voxel rows + 8 pooled voxels, per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const size_t nvox = (size_t)X * Y * Z;
    char* s_st = s_w + NPL * Cout * 128 + wave * (ST_ROWS * ST_PITCH);   // this wave's output staging tile
    const int fr = lane >> 3, fc = lane & 7;
    // persistent over tiles: the weight image and the bias are built once per block
    // 0 <= alpha <= 1 (checked by the entry point): LeakyReLU(v) = max(v, alpha v), two VALU ops instead of compare + multiply
    // + select -- the epilogue's VALU work, not the 64 MFMAs, is what a tile computes (bf16, C2: 1.044 -> 0.968 ms with this
    // and the packed conversions, bit-identical output)
    const float lk = leaky ? alpha : 1.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    const int tzi = bid % ntz; bid /= ntz;
    const int tyi = bid % nty; bid /= nty;
    const int txi = bid % ntx;
    const int b = bid / ntx;
    const int x0 = txi * TX, y0 = tyi * TY, z0 = tzi * TZ;
    __syncthreads();  // previous tile done with s_img
    for (int i = tid; i < HROWS; i += M2_THREADS) {
        const int hx = i / (HY * HZ), hy = (i / HZ) % HY, hz = i % HZ;
        const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
        float a = 0.f, c = 0.f;
        if (gx >= 0 && gx < X && gy >= 0 && gy < Y && gz >= 0 && gz < Z) {
            const size_t o = (size_t)b * nvox + ((size_t)gx * Y + gy) * Z + gz;
            a = src[o];
            c = trg[o];
        }
        s_img[i * 2] = a;
        s_img[i * 2 + 1] = c;
    }
    if (tile == (int)blockIdx.x) {
    for (int i = tid; i < Cout; i += M2_THREADS) s_bias[i] = bias ? bias[i] : 0.f;
    // weight image: row = cout, 64 k (k = tap*2 + ci; k >= 54 zero), chunk swizzle (row >> 1) & 7
    for (int i = tid; i < Cout * 32; i += M2_THREADS) {
        const int co = i % Cout, kp = i / Cout;  // kp = pair index: k = 2 kp, 2 kp + 1 (= tap kp, ci 0/1)
        float f0 = 0.f, f1 = 0.f;
        if (kp < 27) {
            f0 = w[(kp * 2) * Cout + co];
            f1 = w[(kp * 2 + 1) * Cout + co];
        }
        const bf16_t h0 = f32_to_bf16(f0), h1 = f32_to_bf16(f1);
        const int chunk = kp >> 2, within = (kp & 3) * 4;
        const int off = co * 128 + ((chunk ^ ((co >> 1) & 7)) << 4) + within;
        *reinterpret_cast<unsigned*>(s_w + off) = (unsigned)h0 | ((unsigned)h1 << 16);
        if constexpr (X3) {
            const bf16_t l0 = f32_to_bf16(f0 - bf16_to_f32(h0)), l1 = f32_to_bf16(f1 - bf16_to_f32(h1));
            *reinterpret_cast<unsigned*>(s_w + Cout * 128 + off) = (unsigned)l0 | ((unsigned)l1 << 16);
        }
    }
    }
    __syncthreads();
    // X fragments straight from the image patch: for k-step ks a lane holds k = (2 ks + h) * 8 .. + 7 of its voxel, i.e.
    // taps kp = (2 ks + h) * 4 .. + 3 with both channels -- four 8-byte (moving, fixed) pairs of s_img.  (The first version
    // materialised the whole 256 x 64 im2col matrix in LDS with 32 scalar read-convert-write rounds per thread per tile;
    // that pass, not the 64 MFMAs, was what a tile spent its time on besides its 128 KB of output.)
    uint4 xf[NPL][2][4];
#pragma unroll
    for (int vt = 0; vt < 2; ++vt) {
        const int v = wave * 64 + vt * 32 + (lane & 31);
        const int r0 = (vox_x(v) * HY + vox_y(v)) * HZ + vox_z(v);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            unsigned hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // tap index of this element for the two half-waves (compile-time), its patch offset selected by h
                constexpr int dummy = 0; (void)dummy;
                const int kpA = (2 * ks) * 4 + j, kpB = (2 * ks + 1) * 4 + j;
                const int offA = kpA < 27 ? ((kpA / 9) * HY + (kpA / 3) % 3) * HZ + kpA % 3 : -1;
                const int offB = kpB < 27 ? ((kpB / 9) * HY + (kpB / 3) % 3) * HZ + kpB % 3 : -1;
                const int off = h ? offB : offA;
                float2 f = make_float2(0.f, 0.f);
                if (offA >= 0 && offB >= 0) f = *reinterpret_cast<const float2*>(s_img + (r0 + off) * 2);
                else if (offA >= 0 || offB >= 0) {   // only one half-wave has a real tap here (k padding 54..63)
                    const float2 g = *reinterpret_cast<const float2*>(s_img + (r0 + (off >= 0 ? off : 0)) * 2);
                    f = off >= 0 ? g : make_float2(0.f, 0.f);
                }
                const bf16_t h0 = f32_to_bf16(f.x), h1 = f32_to_bf16(f.y);
                hi[j] = (unsigned)h0 | ((unsigned)h1 << 16);
                if constexpr (X3) {
                    const bf16_t l0 = f32_to_bf16(f.x - bf16_to_f32(h0)), l1 = f32_to_bf16(f.y - bf16_to_f32(h1));
                    lo[j] = (unsigned)l0 | ((unsigned)l1 << 16);
                }
            }
            xf[0][vt][ks] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            if constexpr (X3) xf[NPL - 1][vt][ks] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    }
    // flush addressing of this tile: store j of a flush covers the wave's voxel rows 8 j .. 8 j + 7, lane = (row & 7, chunk)
    char* fptr[8];
    unsigned fok = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int vv = wave * 64 + 8 * j + fr;
        const int gx = x0 + vox_x(vv), gy = y0 + vox_y(vv), gz = z0 + vox_z(vv);
        const bool ok = gx < X && gy < Y && gz < Z;
        fok |= (ok ? 1u : 0u) << j;
        const size_t vox = ok ? (size_t)b * nvox + ((size_t)gx * Y + gy) * Z + gz : 0;
        fptr[j] = reinterpret_cast<char*>(out) + vox * Cout * ES_O + fc * 16;
    }
    char* pptr = nullptr;
    bool pok = false;
    if (pool) {   // pooled voxel fr of the wave: x pair = the wave's two x planes, y = fr >> 2, z = fr & 3
        const int X2 = X >> 1, Y2 = Y >> 1, Z2 = Z >> 1;
        const int px = (x0 >> 1) + (wave >> 1), py = (y0 >> 1) + 2 * (wave & 1) + (fr >> 2), pz = (z0 >> 1) + (fr & 3);
        pok = px < X2 && py < Y2 && pz < Z2;
        const size_t vox = pok ? (((size_t)b * X2 + px) * Y2 + py) * Z2 + pz : 0;
        pptr = reinterpret_cast<char*>(pool) + vox * Cout * ES_O + fc * 16;
    }
    for (int n = 0; n < Cout / 32; ++n) {
        f32x16 acc[2];
#pragma unroll
        for (int vt = 0; vt < 2; ++vt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[vt][r] = 0.f;
        const int co_row = n * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int woff = co_row * 128 + (((2 * ks + h) ^ ((co_row >> 1) & 7)) << 4);
            const uint4 wh = *reinterpret_cast<const uint4*>(s_w + woff);
            if constexpr (X3) {
                const uint4 wl = *reinterpret_cast<const uint4*>(s_w + Cout * 128 + woff);
#pragma unroll
                for (int vt = 0; vt < 2; ++vt) {
                    acc[vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl),
                                                                    __builtin_bit_cast(bf16x8, xf[0][vt][ks]), acc[vt], 0, 0, 0);
                    acc[vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh),
                                                                    __builtin_bit_cast(bf16x8, xf[NPL - 1][vt][ks]), acc[vt], 0, 0, 0);
                    acc[vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh),
                                                                    __builtin_bit_cast(bf16x8, xf[0][vt][ks]), acc[vt], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int vt = 0; vt < 2; ++vt)
                    acc[vt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh),
                                                                    __builtin_bit_cast(bf16x8, xf[0][vt][ks]), acc[vt], 0, 0, 0);
            }
        }
        // epilogue: accumulator row = cout n*32 + (r&3) + 8*(r>>2) + 4*h, column = voxel
        float vals2[2][16];
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 bq = *reinterpret_cast<const float4*>(s_bias + n * 32 + 8 * gq + 4 * h);
                const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[vt][gq * 4 + e] + bb[e];
                    v = fmaxf(v, v * lk);
                    vals2[vt][gq * 4 + e] = v;
                }
            }
        }
        // Output through a per-wave LDS staging tile so that every store instruction writes whole 128-B lines: the accumulator
        // layout gives a lane 8 couts (16 B) of ONE voxel per piece, i.e. 32-B runs per voxel and instruction -- stores in that
        // shape alone take the kernel's whole 1.0 ms at 160x160x192x256 (2.8 TB/s; a fill of the same bytes runs at 6.8).  Staged
        // as [voxel row][128 B = 64 bf16 / 32 fp32 couts] (pitch 144 B: conflict-free b128 writes), a flush reads lane =
        // (row & 7, chunk) and stores 8 complete lines per instruction; the pooled voxels (8 per wave) take one more.
        auto stage = [&](int row, const float* vals, bool wr) {   // every lane takes part in the swaps, `wr` lanes write
            char* rp = s_st + row * ST_PITCH;
            if constexpr (OUT_BF16) {
                unsigned pk[4][2];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    pk[gq][0] = pack_bf16x2(vals[gq * 4], vals[gq * 4 + 1]);       // one v_cvt_pk_bf16_f32 per pair
                    pk[gq][1] = pack_bf16x2(vals[gq * 4 + 2], vals[gq * 4 + 3]);
                }
                // pair cout groups (0,1) and (2,3) across the two half-waves -> 8 consecutive couts (16 B) per lane
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        auto rsw = __builtin_amdgcn_permlane32_swap(pk[2 * pr][d], pk[2 * pr + 1][d], false, false);
                        pk[2 * pr][d] = rsw[0];
                        pk[2 * pr + 1][d] = rsw[1];
                    }
                    if (wr)
                        *reinterpret_cast<uint4*>(rp + (((n & 1) * 4 + 2 * pr + h) << 4)) =
                            make_uint4(pk[2 * pr][0], pk[2 * pr][1], pk[2 * pr + 1][0], pk[2 * pr + 1][1]);
                }
            } else {
                if (wr) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<float4*>(rp + ((2 * gq + h) << 4)) =
                            make_float4(vals[gq * 4], vals[gq * 4 + 1], vals[gq * 4 + 2], vals[gq * 4 + 3]);
                }
            }
        };
#pragma unroll
        for (int vt = 0; vt < 2; ++vt) stage(vt * 32 + (lane & 31), vals2[vt], true);
        if (pool) {
            // MaxPooling3D(2) of the activated output (Keras 'valid': floor on odd sizes): x pair = the two accumulator
            // tiles, z pair = lane ^ 1 (DPP quad_perm [1,0,3,2]), y pair = lane ^ 8 (DPP row_ror:8); every lane ends up
            // with the window maximum, the lanes with even y and z stage it (pooled voxel (y >> 1) * 4 + (z >> 1) of the wave)
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = fmaxf(vals2[0][r], vals2[1][r]);
            // v_max_f32 with the DPP operand in the instruction (hipcc keeps update_dpp + max as two).  Inside an asm the
            // compiler pads nothing and a DPP read of a VGPR written by the previous VALU instruction needs 2 wait states: all
            // sixteen z steps, then all sixteen y steps, in ONE block -- every DPP source was written sixteen instructions
            // earlier, so a single s_nop (for the value the compiler computed last) replaces the 32 of the per-value form
#define MMR_PZ(i) "v_max_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define MMR_PY(i) "v_max_f32_dpp %" #i ", %" #i ", %" #i " row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            asm("s_nop 1\n\t"
                MMR_PZ(0) MMR_PZ(1) MMR_PZ(2) MMR_PZ(3) MMR_PZ(4) MMR_PZ(5) MMR_PZ(6) MMR_PZ(7)
                MMR_PZ(8) MMR_PZ(9) MMR_PZ(10) MMR_PZ(11) MMR_PZ(12) MMR_PZ(13) MMR_PZ(14) MMR_PZ(15)
                MMR_PY(0) MMR_PY(1) MMR_PY(2) MMR_PY(3) MMR_PY(4) MMR_PY(5) MMR_PY(6) MMR_PY(7)
                MMR_PY(8) MMR_PY(9) MMR_PY(10) MMR_PY(11) MMR_PY(12) MMR_PY(13) MMR_PY(14) MMR_PY(15)
                : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]),
                  "+v"(pv[8]), "+v"(pv[9]), "+v"(pv[10]), "+v"(pv[11]), "+v"(pv[12]), "+v"(pv[13]), "+v"(pv[14]), "+v"(pv[15]));
#undef MMR_PZ
#undef MMR_PY
            stage(64 + ((lane >> 4) & 1) * 4 + ((lane & 7) >> 1), pv, !(lane & 1) && !(lane & 8));
        }
        if ((n & (G - 1)) == G - 1) {
            const size_t lo = (size_t)(n / G) * 128;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint4 v = *reinterpret_cast<const uint4*>(s_st + (8 * j + fr) * ST_PITCH + fc * 16);
                if ((fok >> j) & 1) *reinterpret_cast<uint4*>(fptr[j] + lo) = v;
            }
            if (pool) {
                const uint4 v = *reinterpret_cast<const uint4*>(s_st + (64 + fr) * ST_PITCH + fc * 16);
                if (pok) *reinterpret_cast<uint4*>(pptr + lo) = v;
            }
        }
    }
    }  // tile loop
}

// ---- flow head (Cout = 3) ---------------------------------------------------------------------------- //
// Padding 3 output channels to an MFMA tile wastes >= 5x.  Instead the 27 taps are folded into the GEMM N
// axis: P[v'][tap*3+co] = sum_ci X[v'][ci] * W[tap][ci][co] is a 1x1x1 contraction with N = 81 (of 96) for
// every voxel v' of a haloed 4x6x10 patch (240 rows = 15 MFMA row tiles exactly), and the 3x3x3 conv is the
// gather-sum out[v][co] = b[co] + sum_tap P[v + off(tap)][tap*3+co] over the patch, done through LDS.
// A fragments come straight from global memory (every lane: one halo voxel, 16 B of channels), weights are
// an LDS image shared by the block's tiles; v_mfma_f32_16x16x32_bf16, fp32 accumulate, fp32 output.
// X3: fp32 input split into bf16 hi/lo on the fly, three MFMAs per product (fp32-grade).
constexpr int FH_TX = 2, FH_TY = 4, FH_TZ = 8;
constexpr int FH_HX = 4, FH_HY = 6, FH_HZ = 10;
constexpr int FH_ROWS = FH_HX * FH_HY * FH_HZ;  // 240
constexpr int FH_THREADS = 256;

template <bool X3>
__global__ void __launch_bounds__(FH_THREADS, 1)
flow_head_kernel(const char* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                 float* __restrict__ out, int B, int X, int Y, int Z, int Cin, int ntx, int nty, int ntz, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nkc = Cin / 8;                       // 16-B k-chunks
    char* sW = smem;                               // [X3 ? 2 : 1][nkc][96][16 B]
    float* sP = reinterpret_cast<float*>(smem + (X3 ? 2 : 1) * nkc * 96 * 16);  // [240][81]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q16 = lane >> 4;
    constexpr int ES = X3 ? 4 : 2;

    // weight image: W'[k = ci][n = tap*3+co]
    for (int i = tid; i < nkc * 96; i += FH_THREADS) {
        const int n = i % 96, kc = i / 96;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float f0 = 0.f, f1 = 0.f;
            if (n < 81) {
                f0 = w[((size_t)(n / 3) * Cin + kc * 8 + 2 * e) * 3 + n % 3];
                f1 = w[((size_t)(n / 3) * Cin + kc * 8 + 2 * e + 1) * 3 + n % 3];
            }
            const bf16_t h0 = f32_to_bf16(f0), h1 = f32_to_bf16(f1);
            hi[e] = (unsigned)h0 | ((unsigned)h1 << 16);
            if constexpr (X3) {
                const bf16_t l0 = f32_to_bf16(f0 - bf16_to_f32(h0)), l1 = f32_to_bf16(f1 - bf16_to_f32(h1));
                lo[e] = (unsigned)l0 | ((unsigned)l1 << 16);
            }
        }
        *reinterpret_cast<uint4*>(sW + (size_t)i * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        if constexpr (X3) *reinterpret_cast<uint4*>(sW + (size_t)(nkc * 96 + i) * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
    const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f, b2 = bias ? bias[2] : 0.f;
    __syncthreads();  // weight image complete

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tzi = t % ntz; t /= ntz;
        const int tyi = t % nty; t /= nty;
        const int txi = t % ntx;
        const int b = t / ntx;
        const int x0 = txi * FH_TX, y0 = tyi * FH_TY, z0 = tzi * FH_TZ;
        // this wave's row tiles: mi = wave + 4 j; per lane the global row pointer of its halo voxel
        const char* rowp[4];
        bool rowok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mi = wave + 4 * j;
            const int hv = mi * 16 + r16;
            const int hx = hv / (FH_HY * FH_HZ), hy = (hv / FH_HZ) % FH_HY, hz = hv % FH_HZ;
            const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
            rowok[j] = (mi < 15) && gx >= 0 && gx < X && gy >= 0 && gy < Y && gz >= 0 && gz < Z;
            const size_t vox = rowok[j] ? ((((size_t)b * X + gx) * Y + gy) * Z + gz) : 0;
            rowp[j] = in + vox * Cin * ES + q16 * (X3 ? 32 : 16);
        }
        f32x4 acc[4][6];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 6; ++n) acc[j][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nsteps = Cin / 32;
        // software pipeline: the raw A loads of k-step s+1 are in flight while step s runs on the matrix cores
        uint4 raw0[4], raw1[4];
        auto load_raw = [&](int s, uint4* r0, uint4* r1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r0[j] = make_uint4(0, 0, 0, 0);
                r1[j] = make_uint4(0, 0, 0, 0);
                if (rowok[j]) {
                    r0[j] = *reinterpret_cast<const uint4*>(rowp[j] + (size_t)s * (X3 ? 128 : 64));
                    if constexpr (X3) r1[j] = *reinterpret_cast<const uint4*>(rowp[j] + (size_t)s * 128 + 16);
                }
            }
        };
        load_raw(0, raw0, raw1);
        for (int s = 0; s < nsteps; ++s) {
            uint4 ah[4], al[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (X3) {
                    const unsigned u[8] = {raw0[j].x, raw0[j].y, raw0[j].z, raw0[j].w, raw1[j].x, raw1[j].y, raw1[j].z, raw1[j].w};
                    unsigned hh[4], ll[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float f0 = __uint_as_float(u[2 * e]), f1 = __uint_as_float(u[2 * e + 1]);
                        const bf16_t h0 = f32_to_bf16(f0), h1 = f32_to_bf16(f1);
                        const bf16_t l0 = f32_to_bf16(f0 - bf16_to_f32(h0)), l1 = f32_to_bf16(f1 - bf16_to_f32(h1));
                        hh[e] = (unsigned)h0 | ((unsigned)h1 << 16);
                        ll[e] = (unsigned)l0 | ((unsigned)l1 << 16);
                    }
                    ah[j] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
                    al[j] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
                } else {
                    ah[j] = raw0[j];
                }
            }
            if (s + 1 < nsteps) load_raw(s + 1, raw0, raw1);
#pragma unroll
            for (int n = 0; n < 6; ++n) {
                const int boff = ((4 * s + q16) * 96 + n * 16 + r16) * 16;
                const uint4 bh = *reinterpret_cast<const uint4*>(sW + boff);
                if constexpr (X3) {
                    const uint4 bl = *reinterpret_cast<const uint4*>(sW + (size_t)nkc * 96 * 16 + boff);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[j]),
                                                                          __builtin_bit_cast(bf16x8, bh), acc[j][n], 0, 0, 0);
                        acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[j]),
                                                                          __builtin_bit_cast(bf16x8, bl), acc[j][n], 0, 0, 0);
                        acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[j]),
                                                                          __builtin_bit_cast(bf16x8, bh), acc[j][n], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[j]),
                                                                          __builtin_bit_cast(bf16x8, bh), acc[j][n], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // previous tile's gather finished reading sP (and the weight image is in place)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mi = wave + 4 * j;
            if (mi < 15) {
#pragma unroll
                for (int n = 0; n < 6; ++n) {
                    const int col = n * 16 + r16;
                    if (col < 81) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) sP[(mi * 16 + q16 * 4 + r) * 81 + col] = acc[j][n][r];
                    }
                }
            }
        }
        __syncthreads();
        if (tid < FH_TX * FH_TY * FH_TZ * 3) {
            const int co = tid % 3, v = tid / 3;
            const int vz = v % FH_TZ, vy = (v / FH_TZ) % FH_TY, vx = v / (FH_TZ * FH_TY);
            const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
            if (gx < X && gy < Y && gz < Z) {
                float a = co == 0 ? b0 : (co == 1 ? b1 : b2);
#pragma unroll
                for (int tap = 0; tap < 27; ++tap) {
                    const int hv = ((vx + tap / 9) * FH_HY + vy + (tap / 3) % 3) * FH_HZ + vz + tap % 3;
                    a += sP[hv * 81 + tap * 3 + co];
                }
                out[((((size_t)b * X + gx) * Y + gy) * Z + gz) * 3 + co] = a;
            }
        }
    }
}

// ---- flow head, marching variant ----------------------------------------------------------------------- //
// Same P-GEMM, but the block walks along x: one input plane of an 8(y) x 16(z) haloed patch = 128 rows = 8 MFMA
// row tiles exactly is contracted per step (P plane [128][81] in LDS, double buffered), and every output thread
// (6 x 14 voxels x 3 channels = 252 of 256 threads) keeps the accumulators of the three output planes that
// input plane touches in registers: plane xp feeds out[xp+1] (dx = 0 taps), out[xp] (dx = 1), out[xp-1] (dx = 2,
// which completes it).  Halo redundancy of the input reads / P-GEMM drops from 3.75x (2x4x8 tiles) to 1.52x
// (+2 planes per x segment); one barrier per plane.
constexpr int MH_TY = 6, MH_TZ = 14, MH_HY = 8, MH_HZ = 16;
constexpr int MH_ROWS = MH_HY * MH_HZ;  // 128
constexpr int MH_THREADS = 256;

// NS > 0: Cin = 32 NS known at compile time (bf16 only): ALL A fragments of a plane (NS loads of 16 B per row tile) are
// issued before the first MFMA, from clamped always-valid addresses, and masked when used.  The generic path (NS = 0)
// loads one k-step ahead behind `if (row in volume)`, which hipcc turns into a branch and a full wait per load: the
// 2 x Cin/32 loads of a plane ran as dependent round trips and the kernel streamed 1.8 TB/s of its 2.5 GB input.
// HY (round 4): y rows of the haloed patch = 2 x the waves of the workgroup.  HY = 16 (bf16, Cin = 256): 14 x 14 outputs per patch
// (halo redundancy 1.31 instead of 1.52), eight waves with two row tiles each -- twice the prefetched bytes in flight per CU --,
// ONE P plane of 256 rows (two barriers per plane), up to two outputs per thread.
template <bool X3, int NS = 0, int HY = MH_HY>
__global__ void __launch_bounds__(HY * 32, 1)
flow_head_march_kernel(const char* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                       float* __restrict__ out, int B, int X, int Y, int Z, int Cin, int nseg, int seglen, int nty, int ntz,
                       int ntiles, int pbufs)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TYO = HY - 2, ROWS = HY * MH_HZ, THREADS = HY * 32;
    constexpr int NOUT = TYO * MH_TZ * 3, OPT = (NOUT + THREADS - 1) / THREADS;   // outputs (voxel, channel) of a patch plane, per thread
    const int nkc = Cin / 8;                       // 16-B k-chunks
    char* sW = smem;                               // [X3 ? 2 : 1][nkc][96][16 B]
    float* sP = reinterpret_cast<float*>(smem + (X3 ? 2 : 1) * nkc * 96 * 16);  // [pbufs][128][81]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q16 = lane >> 4;
    constexpr int ES = X3 ? 4 : 2;

    for (int i = tid; i < nkc * 96; i += THREADS) {  // weight image W'[k = ci][n = tap*3+co], once per block
        const int n = i % 96, kc = i / 96;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float f0 = 0.f, f1 = 0.f;
            if (n < 81) {
                f0 = w[((size_t)(n / 3) * Cin + kc * 8 + 2 * e) * 3 + n % 3];
                f1 = w[((size_t)(n / 3) * Cin + kc * 8 + 2 * e + 1) * 3 + n % 3];
            }
            const bf16_t h0 = f32_to_bf16(f0), h1 = f32_to_bf16(f1);
            hi[e] = (unsigned)h0 | ((unsigned)h1 << 16);
            if constexpr (X3) {
                const bf16_t l0 = f32_to_bf16(f0 - bf16_to_f32(h0)), l1 = f32_to_bf16(f1 - bf16_to_f32(h1));
                lo[e] = (unsigned)l0 | ((unsigned)l1 << 16);
            }
        }
        *reinterpret_cast<uint4*>(sW + (size_t)i * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        if constexpr (X3) *reinterpret_cast<uint4*>(sW + (size_t)(nkc * 96 + i) * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
    // output role of this thread: (voxel, channel) number tid of the patch plane -- and, HY = 16, number tid + THREADS as well
    static_assert(OPT == 1 || OPT == 2, "one or two outputs per thread");
    // HY = 16 (round 5): bank-conflict-free gather.  A ds_read_b32 is served in two groups of 32 lanes over 32 banks; with
    // (voxel, channel) = tid the 32 lanes of a group read row * 81 + tap * 3 + co for ~11 z-neighbouring rows x 3 channels --
    // 2-way conflicts on every read (SQ_LDS_BANK_CONFLICT 29 % of the LDS cycles, profiles/r04k_infer_pmc_sq.json).  Now a
    // 32-lane group is ONE channel of TWO y rows x 14 z (28 lanes, 4 idle): its rows are hz .. hz + 13 and hz + 16 .. hz + 29,
    // distinct mod 32, times the odd pitch 81 -> 28 distinct banks.  21 such (channel, row pair) items over the 16 groups.
    constexpr bool CF = HY == 16;
    const int cf_l = tid & 31, cf_g = tid >> 5;
    const int cf_i1 = cf_g + 16;                                     // second item of groups 0 .. 4
    const int co = CF ? cf_g % 3 : tid % 3, ov = tid / 3;
    const int vz = CF ? cf_l % MH_TZ : ov % MH_TZ, vy = CF ? 2 * (cf_g / 3) + cf_l / MH_TZ : ov / MH_TZ;  // vy < TYO
    const bool outthr = CF ? cf_l < 2 * MH_TZ : tid < NOUT;
    const float bco = bias ? bias[co] : 0.f;
    const int idx1 = (tid + THREADS < NOUT) ? tid + THREADS : 0;
    const int co1 = CF ? cf_i1 % 3 : idx1 % 3, vz1 = CF ? vz : (idx1 / 3) % MH_TZ,
              vy1 = CF ? 2 * (cf_i1 / 3) + cf_l / MH_TZ : (idx1 / 3) / MH_TZ;
    const bool outthr1 = OPT == 2 && (CF ? (cf_l < 2 * MH_TZ && cf_i1 < 3 * (TYO / 2)) : tid + THREADS < NOUT);
    const float bco1 = (OPT == 2 && bias) ? bias[co1] : 0.f;
    const int nsteps = Cin / 32;

    for (int it = 0; it * (int)gridDim.x < ntiles; ++it) {
        int t = xcd_tile((int)blockIdx.x, (int)gridDim.x, it, ntiles);     // y / z neighbours of a segment on one XCD
        if (t < 0) continue;
        const int tzi = t % ntz; t /= ntz;
        const int tyi = t % nty; t /= nty;
        const int seg = t % nseg;
        const int b = t / nseg;
        const int y0 = tyi * TYO, z0 = tzi * MH_TZ;
        const int xs = seg * seglen, xe = (xs + seglen < X) ? xs + seglen : X;
        // this wave's two row tiles = halo y-rows hy = 2 wave + j, lane r16 = halo z
        size_t rowoff[2];
        bool rowok[2];
        unsigned rowmask[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gy = y0 + 2 * wave + j - 1, gz = z0 + r16 - 1;
            rowok[j] = gy >= 0 && gy < Y && gz >= 0 && gz < Z;
            rowmask[j] = rowok[j] ? 0xffffffffu : 0u;
            rowoff[j] = (rowok[j] ? ((size_t)gy * Z + gz) * Cin * ES : 0) + q16 * (X3 ? 32 : 16);
        }
        const int gyo = y0 + vy, gzo = z0 + vz;
        const bool ook = outthr && gyo < Y && gzo < Z;
        float a_prev = 0.f, a_cur = 0.f;
        const int gyo1 = y0 + vy1, gzo1 = z0 + vz1;
        const bool ook1 = outthr1 && gyo1 < Y && gzo1 < Z;
        float a_prev1 = 0.f, a_cur1 = 0.f;
        // NS > 0: the A fragments of plane xp + 1 are loaded while plane xp is multiplied and gathered (native 128-bit vectors,
        // so that the loop-carried registers get no copies behind the loads).  With one 4-wave workgroup per CU (130 KB of
        // LDS) nothing else hides the HBM latency: a plane took 12 k cycles for 1.5 k cycles of MFMA.
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        constexpr int NL = NS > 0 ? NS * (X3 ? 2 : 1) : 1;   // 16-B loads per row and plane (fp32: two per k-step)
        // TWO planes ahead: the loop is unrolled by two over two prefetch buffers (plane xp is consumed from one, which is then
        // reloaded with plane xp + 2, while plane xp + 1 is still in flight in the other).  One plane ahead = 64 KB in flight
        // per CU and 3.9 TB/s of loads at the ~4 us a round trip takes under this load: the kernel was bound by its bytes
        // in flight, not by HBM bandwidth or by its halo (an XCD-contiguous tile order changes nothing).
        // (bf16 only: the narrow fp32x3 inputs run two workgroups per CU, which already doubles the bytes in flight; two planes
        // ahead cost them 5 %)
        constexpr bool DEEP = !X3 && HY == MH_HY;   // HY = 16: eight waves already hold twice the bytes in flight; a second buffer spills (52 B)
        constexpr int AHEAD = DEEP ? 2 : 1;
        typedef u32x4_t PreBuf[NL][2];
        PreBuf preA, preB;
        auto load_plane = [&](PreBuf& pre, int xq) {
            const int xc = xq < 0 ? 0 : (xq >= X ? X - 1 : xq);   // planes outside the volume are never used: any valid address
            const char* pl = in + ((size_t)b * X + xc) * Y * Z * Cin * ES;
#pragma unroll
            for (int s2 = 0; s2 < NL; ++s2)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    pre[s2][j] = *reinterpret_cast<const u32x4_t*>(pl + rowoff[j] +
                                                                   (X3 ? (size_t)(s2 >> 1) * 128 + (s2 & 1) * 16 : (size_t)s2 * 64));
        };
        if constexpr (NS > 0) {
            load_plane(preA, xs - 1);
            if constexpr (DEEP) load_plane(preB, xs);
        }
        auto plane_step = [&](int xp, PreBuf& pre) {
            const int buf = (pbufs == 2) ? ((xp - xs + 1) & 1) : 0;
            float* P = sP + buf * (ROWS * 81);
            const bool inside = xp >= 0 && xp < X;
            if constexpr (NS > 0) {
                if (!inside) load_plane(pre, xp + AHEAD);
            }
            if (inside) {
                const char* plane = in + ((size_t)b * X + xp) * Y * Z * Cin * ES;
                f32x4 acc[2][6];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int n = 0; n < 6; ++n) acc[j][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                uint4 raw0[2], raw1[2];
                auto load_raw = [&](int s, uint4* r0, uint4* r1) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        r0[j] = make_uint4(0, 0, 0, 0);
                        r1[j] = make_uint4(0, 0, 0, 0);
                        if (rowok[j]) {
                            r0[j] = *reinterpret_cast<const uint4*>(plane + rowoff[j] + (size_t)s * (X3 ? 128 : 64));
                            if constexpr (X3) r1[j] = *reinterpret_cast<const uint4*>(plane + rowoff[j] + (size_t)s * 128 + 16);
                        }
                    }
                };
                u32x4_t allraw[NL][2];
                if constexpr (NS > 0) {
#pragma unroll
                    for (int s2 = 0; s2 < NL; ++s2)
#pragma unroll
                        for (int j = 0; j < 2; ++j) allraw[s2][j] = pre[s2][j] & rowmask[j];
                    load_plane(pre, xp + AHEAD);
                } else {
                    load_raw(0, raw0, raw1);
                }
#pragma unroll
                for (int s = 0; s < (NS > 0 ? NS : nsteps); ++s) {
                    uint4 ah[2], al[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (NS > 0 && !X3) {
                            ah[j] = make_uint4(allraw[s][j].x, allraw[s][j].y, allraw[s][j].z, allraw[s][j].w);
                        } else if constexpr (X3) {
                            if constexpr (NS > 0) {
                                const u32x4_t p0 = allraw[2 * s][j], p1 = allraw[2 * s + 1][j];
                                raw0[j] = make_uint4(p0.x, p0.y, p0.z, p0.w);
                                raw1[j] = make_uint4(p1.x, p1.y, p1.z, p1.w);
                            }
                            const unsigned u[8] = {raw0[j].x, raw0[j].y, raw0[j].z, raw0[j].w, raw1[j].x, raw1[j].y, raw1[j].z, raw1[j].w};
                            unsigned hh[4], ll[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float f0 = __uint_as_float(u[2 * e]), f1 = __uint_as_float(u[2 * e + 1]);
                                const bf16_t h0 = f32_to_bf16(f0), h1 = f32_to_bf16(f1);
                                const bf16_t l0 = f32_to_bf16(f0 - bf16_to_f32(h0)), l1 = f32_to_bf16(f1 - bf16_to_f32(h1));
                                hh[e] = (unsigned)h0 | ((unsigned)h1 << 16);
                                ll[e] = (unsigned)l0 | ((unsigned)l1 << 16);
                            }
                            ah[j] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
                            al[j] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
                        } else {
                            ah[j] = raw0[j];
                        }
                    }
                    if constexpr (NS == 0) {
                        if (s + 1 < nsteps) load_raw(s + 1, raw0, raw1);
                    }
#pragma unroll
                    for (int n = 0; n < 6; ++n) {
                        const int boff = ((4 * s + q16) * 96 + n * 16 + r16) * 16;
                        const uint4 bh = *reinterpret_cast<const uint4*>(sW + boff);
                        if constexpr (X3) {
                            const uint4 bl = *reinterpret_cast<const uint4*>(sW + (size_t)nkc * 96 * 16 + boff);
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[j]),
                                                                                  __builtin_bit_cast(bf16x8, bh), acc[j][n], 0, 0, 0);
                                acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[j]),
                                                                                  __builtin_bit_cast(bf16x8, bl), acc[j][n], 0, 0, 0);
                                acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[j]),
                                                                                  __builtin_bit_cast(bf16x8, bh), acc[j][n], 0, 0, 0);
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[j]),
                                                                                  __builtin_bit_cast(bf16x8, bh), acc[j][n], 0, 0, 0);
                        }
                    }
                }
                // C/D layout of the 16x16 tile: lane holds rows q16*4 + r (= halo z), column r16
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int n = 0; n < 6; ++n) {
                        const int col = n * 16 + r16;
                        if (col < 81) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) P[((2 * wave + j) * 16 + q16 * 4 + r) * 81 + col] = acc[j][n][r];
                        }
                    }
            }
            __syncthreads();  // P[buf] complete; the other buffer may still be read by slower waves (not written here)
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if (inside && outthr) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dz = 0; dz < 3; ++dz) {
                        const float* pr = P + ((vy + dy) * MH_HZ + vz + dz) * 81 + (dy * 3 + dz) * 3 + co;
                        c0 += pr[0];
                        c1 += pr[27];
                        c2 += pr[54];
                    }
            }
            const float done = a_prev + c2;  // out plane xp - 1 has now seen its three input planes
            if (ook && xp - 1 >= xs && xp - 1 < xe)
                out[((((size_t)b * X + (xp - 1)) * Y + gyo) * Z + gzo) * 3 + co] = done;
            a_prev = a_cur + c1;
            a_cur = bco + c0;
            if constexpr (OPT == 2) {   // the second output of this thread, the same way
                float d0 = 0.f, d1 = 0.f, d2 = 0.f;
                if (inside && outthr1) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dz = 0; dz < 3; ++dz) {
                            const float* pr = P + ((vy1 + dy) * MH_HZ + vz1 + dz) * 81 + (dy * 3 + dz) * 3 + co1;
                            d0 += pr[0];
                            d1 += pr[27];
                            d2 += pr[54];
                        }
                }
                const float done1 = a_prev1 + d2;
                if (ook1 && xp - 1 >= xs && xp - 1 < xe)
                    out[((((size_t)b * X + (xp - 1)) * Y + gyo1) * Z + gzo1) * 3 + co1] = done1;
                a_prev1 = a_cur1 + d1;
                a_cur1 = bco1 + d0;
            }
            if (pbufs == 1) __syncthreads();  // single P buffer (wide fp32x3 inputs): gather done before the next plane lands
        };
        if constexpr (DEEP) {
            for (int xp = xs - 1; xp <= xe; xp += 2) {
                plane_step(xp, preA);
                if (xp + 1 <= xe) plane_step(xp + 1, preB);
            }
        } else {
            for (int xp = xs - 1; xp <= xe; ++xp) plane_step(xp, preA);
        }
        __syncthreads();  // the next tile's first plane reuses buffer 0/1
    }
}

// ---- MaxPooling3D(2), 16 B per lane -------------------------------------- //
template <int DT>
__global__ void __launch_bounds__(256)
maxpool_kernel(const char* __restrict__ in, char* __restrict__ out, int B, int X, int Y, int Z, int C)
{
    constexpr int ES = Elt<DT>::size;
    constexpr int EPC = 16 / ES;
    const int Xo = X / 2, Yo = Y / 2, Zo = Z / 2;
    const int cch = C / EPC;  // 16-B chunks per voxel
    const int64_t total = (int64_t)B * Xo * Yo * Zo * cch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int ch = (int)(r % cch); r /= cch;
        const int z = (int)(r % Zo); r /= Zo;
        const int y = (int)(r % Yo); r /= Yo;
        const int x = (int)(r % Xo);
        const int b = (int)(r / Xo);
        float m[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int gx = 2 * x + (k >> 2), gy = 2 * y + ((k >> 1) & 1), gz = 2 * z + (k & 1);
            const uint4 v = *reinterpret_cast<const uint4*>(
                in + ((((size_t)b * X + gx) * Y + gy) * Z + gz) * C * ES + (size_t)ch * 16);
            if (DT == MMR_DT_BF16) {
                const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m[2 * e] = fmaxf(m[2 * e], __uint_as_float(u[e] << 16));
                    m[2 * e + 1] = fmaxf(m[2 * e + 1], __uint_as_float(u[e] & 0xffff0000u));
                }
            } else {
                m[0] = fmaxf(m[0], __uint_as_float(v.x));
                m[1] = fmaxf(m[1], __uint_as_float(v.y));
                m[2] = fmaxf(m[2], __uint_as_float(v.z));
                m[3] = fmaxf(m[3], __uint_as_float(v.w));
            }
        }
        uint4 o;
        if (DT == MMR_DT_BF16) {
            unsigned u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                u[e] = (__float_as_uint(m[2 * e]) >> 16) | (__float_as_uint(m[2 * e + 1]) & 0xffff0000u);
            o = make_uint4(u[0], u[1], u[2], u[3]);
        } else {
            o = make_uint4(__float_as_uint(m[0]), __float_as_uint(m[1]), __float_as_uint(m[2]), __float_as_uint(m[3]));
        }
        *reinterpret_cast<uint4*>(out + (size_t)i * 16) = o;
    }
}

}  // namespace mmr

using namespace mmr;

extern "C" int64_t mmr_conv3d_k3_packed_bytes(int Cin, int Cout, int dtype)
{
    if (Cin < 1 || Cout < 1) return MMR_EINVAL;
    const int kc = (dtype == MMR_DT_BF16) ? 64 : 32;
    if (dtype < 0 || dtype > MMR_DT_F32X1) return MMR_EINVAL;
    if (Cin % kc) return MMR_EINVAL;
    const int BN = conv_bn(Cout);
    const int nt = (Cout + BN - 1) / BN;
    return (int64_t)nt * (Cin / kc) * 27 * BN * 128;
}

extern "C" int mmr_conv3d_k3_pack(const float* w_keras, void* w_packed, int Cin, int Cout, int dtype,
                                  int transpose_flip, void* stream)
{
    const int64_t bytes = mmr_conv3d_k3_packed_bytes(Cin, Cout, dtype);
    if (!w_keras || !w_packed || bytes < 0) return MMR_EINVAL;
    const int BN = conv_bn(Cout);
    const int nt = (Cout + BN - 1) / BN;
    const int grid = stream_grid(bytes / 16, 256);
    if (dtype == MMR_DT_BF16)
        hipLaunchKernelGGL(pack_kernel<MMR_DT_BF16>, dim3(grid), dim3(256), 0, as_stream(stream), w_keras,
                           (char*)w_packed, Cin, Cout, BN, nt, transpose_flip, 0);
    else if (dtype == MMR_DT_F32X3 || dtype == MMR_DT_F32X1)
        hipLaunchKernelGGL(pack_kernel<MMR_DT_F32X3>, dim3(grid), dim3(256), 0, as_stream(stream), w_keras,
                           (char*)w_packed, Cin, Cout, BN, nt, transpose_flip, 0);
    else
        hipLaunchKernelGGL(pack_kernel<MMR_DT_F32>, dim3(grid), dim3(256), 0, as_stream(stream), w_keras,
                           (char*)w_packed, Cin, Cout, BN, nt, transpose_flip, 0);
    return check_launch();
}

// Scratch for the split-K path of small launches (0 when the launch fills the chip on its own): S partial output
// tensors in fp32.  Passing ws == NULL to mmr_conv3d_k3_fwd_ws (or calling mmr_conv3d_k3_fwd) disables the split.
extern "C" int64_t mmr_conv3d_k3_ksplit_ws_bytes(int B, int X, int Y, int Z, int Cin, int Cout, int dtype)
{
    if (B < 1 || X < 1 || Y < 1 || Z < 1 || Cin < 1 || Cout < 1 || dtype < 0 || dtype > MMR_DT_F32X1) return MMR_EINVAL;
    const int BN = conv_bn(Cout), nt = (Cout + BN - 1) / BN;
    const int txt = (BN == 256) ? 4 : 8;
    const int64_t nblk = (int64_t)B * ((X + txt - 1) / txt) * ((Y + TY - 1) / TY) * ((Z + TZ - 1) / TZ);
    const int G = (Cin / ((dtype == MMR_DT_BF16) ? 64 : 32)) * 27;
    const int S = conv_ksplit(nblk, nt, G);
    if (S > 1) return (int64_t)S * B * X * Y * Z * Cout * (int64_t)sizeof(float);
    // tail split of a launch that leaves a partial round of workgroups (conv_tail_plan): compact partials of the tail tiles
    int R = 0, St = 0;
    if (conv_uses_m16(dtype, BN) && BN >= 64 && (Cout & 15) == 0 && conv_tail_plan(nblk, nt, G, &R, &St)) {
        const int gs = (G + St - 1) / St, nz = (G + gs - 1) / gs;
        return (int64_t)nz * R * (txt * TY * TZ) * Cout * (int64_t)sizeof(float);
    }
    return 0;
}

extern "C" int mmr_conv3d_k3_fwd_ws(const void* in0, int C0, int up0, const void* in1, int C1, const void* w_packed,
                                    const float* bias, void* out, void* pool_out, int B, int X, int Y, int Z, int Cout,
                                    int leaky, float alpha, int dtype, int out_f32, void* ws, void* stream);

extern "C" int mmr_conv3d_k3_fwd(const void* in0, int C0, int up0, const void* in1, int C1, const void* w_packed,
                                 const float* bias, void* out, void* pool_out, int B, int X, int Y, int Z, int Cout,
                                 int leaky, float alpha, int dtype, int out_f32, void* stream)
{
    return mmr_conv3d_k3_fwd_ws(in0, C0, up0, in1, C1, w_packed, bias, out, pool_out, B, X, Y, Z, Cout, leaky, alpha, dtype,
                                out_f32, nullptr, stream);
}

extern "C" int mmr_conv3d_k3_fwd_ws(const void* in0, int C0, int up0, const void* in1, int C1, const void* w_packed,
                                    const float* bias, void* out, void* pool_out, int B, int X, int Y, int Z, int Cout,
                                    int leaky, float alpha, int dtype, int out_f32, void* ws, void* stream)
{
    if (!in0 || !w_packed || !out || B < 1 || X < 1 || Y < 1 || Z < 1 || Cout < 1 || C0 < 1 || C1 < 0) return MMR_EINVAL;
    if (dtype < 0 || dtype > MMR_DT_F32X1) return MMR_EINVAL;
    if (C1 > 0 && !in1) return MMR_EINVAL;
    if (pool_out) return MMR_EUNSUPPORTED;  // fused pooling: planned; use mmr_maxpool3d2_fwd
    const int kc = (dtype == MMR_DT_BF16) ? 64 : 32;
    if (C0 % kc || C1 % kc) return MMR_EINVAL;
    if (up0 && ((X | Y | Z) & 1)) return MMR_EINVAL;
    ConvParams p;
    p.in0 = (const char*)in0; p.in1 = (const char*)in1; p.wp = (const char*)w_packed; p.bias = bias;
    p.out = (char*)out;
    p.B = B; p.X = X; p.Y = Y; p.Z = Z; p.C0 = C0; p.C1 = C1; p.up0 = up0; p.Cout = Cout;
    p.leaky = leaky; p.alpha = alpha; p.out_f32 = out_f32; p.ymask = nullptr; p.part = nullptr;
    p.kpart = (float*)ws; p.gsplit = 0; p.out1 = nullptr; p.csplit = 0;
    p.ntx = (X + TX - 1) / TX; p.nty = (Y + TY - 1) / TY; p.ntz = (Z + TZ - 1) / TZ;
    if (dtype == MMR_DT_BF16) return dispatch_conv<MMR_DT_BF16>(p, as_stream(stream));
    if (dtype == MMR_DT_F32X3) return dispatch_conv<MMR_DT_F32X3>(p, as_stream(stream));
    if (dtype == MMR_DT_F32X1) return dispatch_conv<MMR_DT_F32X1>(p, as_stream(stream));
    return dispatch_conv<MMR_DT_F32>(p, as_stream(stream));
}

// ---- folded upsampling (decoder layers: conv over concat([UpSampling3D(2)(x) | skip]); see CV_UPFOLD) ---- //
extern "C" int64_t mmr_conv3d_k3_upfold_packed_bytes(int C0, int Cout, int dtype)
{
    if (dtype != MMR_DT_BF16 && dtype != MMR_DT_F32X3) return MMR_EUNSUPPORTED;
    const int kc = (dtype == MMR_DT_BF16) ? 64 : 32;
    if (C0 < kc || C0 % kc || Cout < 64 || Cout % 64) return MMR_EINVAL;
    const int BN = conv_bn(Cout);
    return (int64_t)8 * (Cout / BN) * (C0 / kc) * 8 * BN * 128;
}

// w_up_keras: [27][C0][Cout] fp32 = the first C0 input channels of the layer's Keras kernel (contiguous)
extern "C" int mmr_conv3d_k3_upfold_pack(const float* w_up_keras, void* w_packed, int C0, int Cout, int dtype, void* stream)
{
    const int64_t bytes = mmr_conv3d_k3_upfold_packed_bytes(C0, Cout, dtype);
    if (bytes < 0) return (int)bytes;
    if (!w_up_keras || !w_packed) return MMR_EINVAL;
    const int BN = conv_bn(Cout);
    const int grid = stream_grid(bytes / 16, 256);
    if (dtype == MMR_DT_BF16)
        hipLaunchKernelGGL(pack_upfold_kernel<MMR_DT_BF16>, dim3(grid), dim3(256), 0, as_stream(stream), w_up_keras,
                           (char*)w_packed, C0, Cout, BN, Cout / BN);
    else
        hipLaunchKernelGGL(pack_upfold_kernel<MMR_DT_F32X3>, dim3(grid), dim3(256), 0, as_stream(stream), w_up_keras,
                           (char*)w_packed, C0, Cout, BN, Cout / BN);
    return check_launch();
}

// partial [B,2X2,2Y2,2Z2,Cout] = conv3x3x3(UpSampling3D(2)(in_low)) restricted to the C0 upsampled channels, no bias.
// in_low [B,X2,Y2,Z2,C0] bf16 (MMR_DT_BF16) or fp32 (MMR_DT_F32X3).  Every element of `partial` is written.
// partial_half != 0 (MMR_DT_BF16 only): `partial` holds IEEE half values saturated to +-65504 instead of fp32.
extern "C" int mmr_conv3d_k3_upfold_fwd(const void* in_low, int C0, const void* w_packed, void* partial, int partial_half,
                                        int B, int X2, int Y2, int Z2, int Cout, int dtype, void* stream)
{
    if (partial_half && dtype != MMR_DT_BF16) return MMR_EUNSUPPORTED;
    if (!in_low || !w_packed || !partial || B < 1 || X2 < 1 || Y2 < 1 || Z2 < 1) return MMR_EINVAL;
    if (mmr_conv3d_k3_upfold_packed_bytes(C0, Cout, dtype) < 0) return (int)mmr_conv3d_k3_upfold_packed_bytes(C0, Cout, dtype);
    ConvParams p;
    p.in0 = (const char*)in_low; p.in1 = nullptr; p.wp = (const char*)w_packed; p.bias = nullptr;
    p.out = (char*)partial;
    p.B = B; p.X = X2; p.Y = Y2; p.Z = Z2; p.C0 = C0; p.C1 = 0; p.up0 = 0; p.Cout = Cout;
    p.leaky = 0; p.alpha = 0.f; p.out_f32 = 1; p.ymask = nullptr; p.part = nullptr;
    p.kpart = nullptr; p.gsplit = 0; p.out1 = nullptr; p.csplit = 0;
    p.ntx = (X2 + TX - 1) / TX; p.nty = (Y2 + TY - 1) / TY; p.ntz = (Z2 + TZ - 1) / TZ;
    if (dtype == MMR_DT_BF16 && partial_half) return dispatch_conv_fold<MMR_DT_BF16, CV_UPFOLD | CV_PART16>(p, as_stream(stream));
    if (dtype == MMR_DT_BF16) return dispatch_conv_fold<MMR_DT_BF16, CV_UPFOLD>(p, as_stream(stream));
    return dispatch_conv_fold<MMR_DT_F32X3, CV_UPFOLD>(p, as_stream(stream));
}

// out = act(cinit + conv3x3x3(in) + bias): the skip half of a folded layer (cinit = mmr_conv3d_k3_upfold_fwd's partial),
// otherwise mmr_conv3d_k3_fwd without the concat / split-K options.
extern "C" int mmr_conv3d_k3_fwd_init(const void* in, int Cin, const void* w_packed, const float* bias, const void* cinit,
                                      int cinit_half, void* out, int B, int X, int Y, int Z, int Cout, int leaky, float alpha,
                                      int dtype, int out_f32, void* ws, void* stream)
{
    if (cinit_half && dtype != MMR_DT_BF16) return MMR_EUNSUPPORTED;
    if (!in || !w_packed || !out || !cinit || B < 1 || X < 1 || Y < 1 || Z < 1) return MMR_EINVAL;
    if (dtype != MMR_DT_BF16 && dtype != MMR_DT_F32X3) return MMR_EUNSUPPORTED;
    const int kc = (dtype == MMR_DT_BF16) ? 64 : 32;
    if (Cin < kc || Cin % kc || Cout < 64 || Cout % 64) return MMR_EINVAL;
    ConvParams p;
    p.in0 = (const char*)in; p.in1 = nullptr; p.wp = (const char*)w_packed; p.bias = bias;
    p.out = (char*)out;
    p.B = B; p.X = X; p.Y = Y; p.Z = Z; p.C0 = Cin; p.C1 = 0; p.up0 = 0; p.Cout = Cout;
    p.leaky = leaky; p.alpha = alpha; p.out_f32 = out_f32; p.ymask = nullptr; p.part = nullptr;
    p.kpart = (float*)ws; p.gsplit = 0; p.out1 = nullptr; p.csplit = 0; p.cinit = (const float*)cinit;
    p.ntx = (X + TX - 1) / TX; p.nty = (Y + TY - 1) / TY; p.ntz = (Z + TZ - 1) / TZ;
    if (dtype == MMR_DT_BF16 && cinit_half) return dispatch_conv_fold<MMR_DT_BF16, CV_CINIT | CV_PART16>(p, as_stream(stream));
    if (dtype == MMR_DT_BF16) return dispatch_conv_fold<MMR_DT_BF16, CV_CINIT>(p, as_stream(stream));
    return dispatch_conv_fold<MMR_DT_F32X3, CV_CINIT>(p, as_stream(stream));
}

// ---- data gradient of the folded half (see CV_DGFOLD) ---- //
extern "C" int64_t mmr_conv3d_k3_dgrad_upfold_packed_bytes(int Cz, int C0, int dtype)
{
    if (dtype != MMR_DT_F32X3 && dtype != MMR_DT_F32X1) return MMR_EUNSUPPORTED;
    if (Cz < 32 || Cz % 32 || C0 < 64 || C0 % 64) return MMR_EINVAL;
    const int BN = conv_bn(C0);
    return (int64_t)(C0 / BN) * 8 * (Cz / 32) * 8 * BN * 128;
}

// w_up_keras: [27][C0][Cz] fp32 = the first C0 input channels of the layer's FORWARD Keras kernel (contiguous)
extern "C" int mmr_conv3d_k3_dgrad_upfold_pack(const float* w_up_keras, void* w_packed, int C0, int Cz, int dtype, void* stream)
{
    const int64_t bytes = mmr_conv3d_k3_dgrad_upfold_packed_bytes(Cz, C0, dtype);
    if (bytes < 0) return (int)bytes;
    if (!w_up_keras || !w_packed) return MMR_EINVAL;
    const int BN = conv_bn(C0);
    hipLaunchKernelGGL(pack_dgfold_kernel<MMR_DT_F32X3>, dim3(stream_grid(bytes / 16, 256)), dim3(256), 0, as_stream(stream),
                       w_up_keras, (char*)w_packed, C0, Cz, BN, C0 / BN);
    return check_launch();
}

// Bytes of the weight image job (kind, rows, cols) produces (the *_packed_bytes of the matching single-image entry point).
extern "C" int64_t mmr_conv3d_k3_pack_job_bytes(int kind, int rows, int cols, int dtype)
{
    switch (kind) {
    case MMR_PACK_FWD: return mmr_conv3d_k3_packed_bytes(rows, cols, dtype);
    case MMR_PACK_DGRAD: return mmr_conv3d_k3_packed_bytes(cols, rows, dtype);
    case MMR_PACK_UPFOLD: return mmr_conv3d_k3_upfold_packed_bytes(rows, cols, dtype);
    case MMR_PACK_DGFOLD: return mmr_conv3d_k3_dgrad_upfold_packed_bytes(cols, rows, dtype);
    }
    return MMR_EINVAL;
}

template <int DT>
static int pack_batch_launch(const MmrPackJob* jobs, int njobs, int dtype, hipStream_t st)
{
    for (int j0 = 0; j0 < njobs; j0 += PACK_BATCH_MAX) {
        PackBatch pb;
        pb.n = njobs - j0 < PACK_BATCH_MAX ? njobs - j0 : PACK_BATCH_MAX;
        pb.total = 0;
        for (int k = 0; k < pb.n; ++k) {
            const MmrPackJob& q = jobs[j0 + k];
            PackJobDev& d = pb.j[k];
            const int64_t bytes = mmr_conv3d_k3_pack_job_bytes(q.kind, q.rows, q.cols, dtype);
            d.w = q.w; d.out = (char*)q.out; d.first = pb.total; d.kind = q.kind; d.ld = q.rows_total; d.off = q.row_off;
            d.a = q.rows; d.b = q.cols;
            const int ncols = (q.kind == MMR_PACK_FWD || q.kind == MMR_PACK_UPFOLD) ? q.cols : q.rows;   // the image's N axis
            const int nk = (q.kind == MMR_PACK_FWD || q.kind == MMR_PACK_UPFOLD) ? q.rows : q.cols;      // its contraction axis
            d.BN = conv_bn(ncols);
            d.ntiles = (ncols + d.BN - 1) / d.BN;
            d.nsl = nk / Elt<DT>::kc;
            pb.total += bytes / 16;
        }
        hipLaunchKernelGGL(pack_batch_kernel<DT>, dim3(stream_grid(pb.total, 256)), dim3(256), 0, st, pb);
    }
    return check_launch();
}

// Several weight images in one launch: what mmr_conv3d_k3_pack (kinds FWD, DGRAD = transpose_flip), mmr_conv3d_k3_upfold_pack
// and mmr_conv3d_k3_dgrad_upfold_pack write one at a time, bit for bit, reading channel slices of the Keras kernels in place.
// `jobs` is a HOST array (copied into the launch).  All images of a call share `dtype`.
extern "C" int mmr_conv3d_k3_pack_batch(const MmrPackJob* jobs, int njobs, int dtype, void* stream)
{
    if (!jobs || njobs < 1 || dtype < 0 || dtype > MMR_DT_F32X1) return MMR_EINVAL;
    for (int k = 0; k < njobs; ++k) {
        const MmrPackJob& q = jobs[k];
        if (!q.w || !q.out || q.rows < 1 || q.cols < 1 || q.row_off < 0 || q.row_off + q.rows > q.rows_total) return MMR_EINVAL;
        const int64_t bytes = mmr_conv3d_k3_pack_job_bytes(q.kind, q.rows, q.cols, dtype);
        if (bytes < 0) return (int)bytes;
    }
    if (dtype == MMR_DT_BF16) return pack_batch_launch<MMR_DT_BF16>(jobs, njobs, dtype, as_stream(stream));
    if (dtype == MMR_DT_F32) return pack_batch_launch<MMR_DT_F32>(jobs, njobs, dtype, as_stream(stream));
    return pack_batch_launch<MMR_DT_F32X3>(jobs, njobs, dtype, as_stream(stream));
}

extern "C" int64_t mmr_conv3d_k3_dgrad_upfold_ws_bytes(int B, int X2, int Y2, int Z2, int C0)
{
    return (int64_t)B * ((X2 + TX - 1) / TX) * ((Y2 + TY - 1) / TY) * ((Z2 + TZ - 1) / TZ) * C0 * (int64_t)sizeof(double);
}

// d x_low [B,X2,Y2,Z2,C0] fp32 = gradient of conv3x3x3(concat([UpSampling3D(2)(x_low) | skip])) w.r.t. x_low given
// dz [B,2X2,2Y2,2Z2,Cz]: what mmr_conv3d_k3_dgrad_split + the 2x2x2 pooling of mmr_upcat_bwd_masked_f32 produce for the
// upsampled channels.  ymask != NULL: multiplied by LeakyReLU'(ymask) (ymask = x_low itself, the activated output of the
// layer that produced it) and dbias (+)= the column sums of the result, as in mmr_conv3d_k3_dgrad_masked.
extern "C" int mmr_conv3d_k3_dgrad_upfold(const void* dz, int Cz, const void* w_packed, float* out, int B, int X2, int Y2,
                                          int Z2, int C0, const float* ymask, float alpha, float* dbias, void* ws,
                                          int accumulate, int dtype, void* stream)
{
    if (!dz || !w_packed || !out || B < 1 || X2 < 1 || Y2 < 1 || Z2 < 1) return MMR_EINVAL;
    if (mmr_conv3d_k3_dgrad_upfold_packed_bytes(Cz, C0, dtype) < 0) return (int)mmr_conv3d_k3_dgrad_upfold_packed_bytes(Cz, C0, dtype);
    if (ymask && (!dbias || !ws)) return MMR_EINVAL;
    ConvParams p;
    p.in0 = (const char*)dz; p.in1 = nullptr; p.wp = (const char*)w_packed; p.bias = nullptr;
    p.out = (char*)out;
    p.B = B; p.X = X2; p.Y = Y2; p.Z = Z2; p.C0 = Cz; p.C1 = 0; p.up0 = 0; p.Cout = C0;
    p.leaky = 0; p.alpha = alpha; p.out_f32 = 1; p.ymask = ymask; p.part = ymask ? (double*)ws : nullptr;
    p.kpart = nullptr; p.gsplit = 0; p.out1 = nullptr; p.csplit = 0;
    p.ntx = (X2 + TX - 1) / TX; p.nty = (Y2 + TY - 1) / TY; p.ntz = (Z2 + TZ - 1) / TZ;
    int64_t nblk = 0;
    int rc;
    if (dtype == MMR_DT_F32X3) rc = dispatch_conv_fold<MMR_DT_F32X3, CV_DGFOLD>(p, as_stream(stream), &nblk);
    else rc = dispatch_conv_fold<MMR_DT_F32X1, CV_DGFOLD>(p, as_stream(stream), &nblk);
    if (rc != MMR_OK || !ymask) return rc;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(C0), dim3(64), 0, as_stream(stream), (const double*)ws, dbias, C0, (int)nblk,
                       accumulate);
    return check_launch();
}

extern "C" int64_t mmr_conv3d_k3_dgrad_masked_ws_bytes(int B, int X, int Y, int Z, int Cout)
{
    // one row of Cout doubles per spatial block; the narrowest M tile (4x8x8) bounds the block count
    return (int64_t)B * ((X + TX - 1) / TX) * ((Y + TY - 1) / TY) * ((Z + TZ - 1) / TZ) * Cout * (int64_t)sizeof(double);
}

// Data gradient of a k3 conv (w_packed = transposed / flipped weights, no bias) whose result is at once pushed
// through the LeakyReLU backward of the layer that produced the conv's input: out = conv(in) * (ymask < 0 ? alpha : 1),
// dbias (+)= sum over voxels of out.  ymask = that layer's activated output [B,X,Y,Z,Cout] fp32.
static int dgrad_masked_impl(const void* in0, int C0, const void* w_packed, float* out, int B, int X, int Y, int Z, int Cout,
                             const float* ymask, float alpha, float* dbias, void* ws, int accumulate, int dtype,
                             const float* dpool, void* stream);

extern "C" int mmr_conv3d_k3_dgrad_masked(const void* in0, int C0, const void* w_packed, float* out, int B, int X, int Y,
                                          int Z, int Cout, const float* ymask, float alpha, float* dbias, void* ws,
                                          int accumulate, int dtype, void* stream)
{
    return dgrad_masked_impl(in0, C0, w_packed, out, B, X, Y, Z, Cout, ymask, alpha, dbias, ws, accumulate, dtype, nullptr, stream);
}

// The same with the gradient that reaches ymask's tensor y through MaxPooling3D(2) folded in: out = (conv(in0) + route(dpool)) *
// LeakyReLU'(y), dbias (+)= column sums; dpool [B, X/2, Y/2, Z/2, Cout] is routed to the first maximum of every 2x2x2 window of
// y (window order x, y, z).  MMR_DT_F32X3 / F32X1, Cout % 64 == 0 (but not % 128), even X, Y, Z; else MMR_EUNSUPPORTED.
extern "C" int mmr_conv3d_k3_dgrad_masked_pool(const void* in0, int C0, const void* w_packed, float* out, int B, int X, int Y,
                                               int Z, int Cout, const float* ymask, float alpha, float* dbias, void* ws,
                                               int accumulate, int dtype, const float* dpool, void* stream)
{
    if (!dpool) return MMR_EINVAL;
    if ((dtype != MMR_DT_F32X3 && dtype != MMR_DT_F32X1) || conv_bn(Cout) != 64 || ((X | Y | Z) & 1)) return MMR_EUNSUPPORTED;
    return dgrad_masked_impl(in0, C0, w_packed, out, B, X, Y, Z, Cout, ymask, alpha, dbias, ws, accumulate, dtype, dpool, stream);
}

static int dgrad_masked_impl(const void* in0, int C0, const void* w_packed, float* out, int B, int X, int Y, int Z, int Cout,
                             const float* ymask, float alpha, float* dbias, void* ws, int accumulate, int dtype,
                             const float* dpool, void* stream)
{
    if (!in0 || !w_packed || !out || !ymask || !dbias || !ws || B < 1 || X < 1 || Y < 1 || Z < 1 || Cout < 1 || C0 < 1)
        return MMR_EINVAL;
    if (dtype != MMR_DT_F32 && dtype != MMR_DT_F32X3 && dtype != MMR_DT_F32X1) return MMR_EINVAL;
    if (C0 % 32) return MMR_EINVAL;
    ConvParams p;
    p.dpool = dpool;
    p.in0 = (const char*)in0; p.in1 = nullptr; p.wp = (const char*)w_packed; p.bias = nullptr;
    p.out = (char*)out;
    p.B = B; p.X = X; p.Y = Y; p.Z = Z; p.C0 = C0; p.C1 = 0; p.up0 = 0; p.Cout = Cout;
    p.leaky = 0; p.alpha = alpha; p.out_f32 = 1; p.ymask = ymask; p.part = (double*)ws;
    p.kpart = nullptr; p.gsplit = 0; p.out1 = nullptr; p.csplit = 0;
    p.ntx = (X + TX - 1) / TX; p.nty = (Y + TY - 1) / TY; p.ntz = (Z + TZ - 1) / TZ;
    int64_t nblk = 0;
    int rc;
    if (dtype == MMR_DT_F32X3) rc = dispatch_conv<MMR_DT_F32X3>(p, as_stream(stream), &nblk);
    else if (dtype == MMR_DT_F32X1) rc = dispatch_conv<MMR_DT_F32X1>(p, as_stream(stream), &nblk);
    else rc = dispatch_conv<MMR_DT_F32>(p, as_stream(stream), &nblk);
    if (rc != MMR_OK) return rc;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(Cout), dim3(64), 0, as_stream(stream), (const double*)ws, dbias, Cout,
                       (int)nblk, accumulate);
    return check_launch();
}

extern "C" int64_t mmr_conv3d_k3_dgrad_split_ws_bytes(int B, int X, int Y, int Z, int C1)
{
    return (int64_t)B * ((X + TX - 1) / TX) * ((Y + TY - 1) / TY) * ((Z + TZ - 1) / TZ) * C1 * (int64_t)sizeof(double);
}

// Data gradient of a conv whose input was concat([up2(in0) | in0, in1]): the first C0 gradient channels are stored
// compactly to d0 [B,X,Y,Z,C0] (to be pool-summed by mmr_upcat_bwd_masked_f32 with C1 = 0), the last C1 channels
// directly to d1 [B,X,Y,Z,C1], multiplied by LeakyReLU'(y1) with dbias1 (+)= their column sums when y1 is given.
// The full concatenated gradient is never materialised.  16x16x32 kernels only (fp32x3 / x1, Cout = C0 + C1 a multiple
// of 64, C0 and C1 multiples of 16), else MMR_EUNSUPPORTED.
extern "C" int mmr_conv3d_k3_dgrad_split(const void* dz, int Cz, const void* w_packed, float* d0, float* d1, int B, int X,
                                         int Y, int Z, int C0, int C1, const float* y1, float alpha, float* dbias1,
                                         void* ws, int accumulate, int dtype, void* stream)
{
    if (!dz || !w_packed || !d0 || !d1 || B < 1 || X < 1 || Y < 1 || Z < 1 || C0 < 1 || C1 < 1 || Cz < 1) return MMR_EINVAL;
    if (y1 && (!dbias1 || !ws)) return MMR_EINVAL;
    if (dtype != MMR_DT_F32X3 && dtype != MMR_DT_F32X1) return MMR_EUNSUPPORTED;
    const int Cout = C0 + C1;
    if (Cz % 32 || !conv_uses_m16(dtype, conv_bn(Cout)) || (C0 & 15) || (C1 & 15)) return MMR_EUNSUPPORTED;
    ConvParams p;
    p.in0 = (const char*)dz; p.in1 = nullptr; p.wp = (const char*)w_packed; p.bias = nullptr;
    p.out = (char*)d0; p.out1 = (char*)d1; p.csplit = C0;
    p.B = B; p.X = X; p.Y = Y; p.Z = Z; p.C0 = Cz; p.C1 = 0; p.up0 = 0; p.Cout = Cout;
    p.leaky = 0; p.alpha = alpha; p.out_f32 = 1; p.ymask = y1; p.part = (double*)ws;
    p.kpart = nullptr; p.gsplit = 0;
    p.ntx = (X + TX - 1) / TX; p.nty = (Y + TY - 1) / TY; p.ntz = (Z + TZ - 1) / TZ;
    int64_t nblk = 0;
    const int rc = (dtype == MMR_DT_F32X3) ? dispatch_conv<MMR_DT_F32X3>(p, as_stream(stream), &nblk)
                                           : dispatch_conv<MMR_DT_F32X1>(p, as_stream(stream), &nblk);
    if (rc != MMR_OK || !y1) return rc;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(C1), dim3(64), 0, as_stream(stream), (const double*)ws, dbias1, C1,
                       (int)nblk, accumulate);
    return check_launch();
}

template <bool X3, bool OUT_BF16>
static int launch_cin2_mfma(const float* src, const float* trg, const float* w, const float* bias, void* out, void* pool,
                            int B, int X, int Y, int Z, int Cout, int leaky, float alpha, hipStream_t st)
{
    const int npl = X3 ? 2 : 1;
    const int lds = HROWS * 8 + 2048 + npl * Cout * 128 + 4 * 72 * 144;   // + the four waves' output staging tiles
    if (lds > 160 * 1024) return MMR_EUNSUPPORTED;
    static bool attr_set = false;
    auto kern = conv3d_cin2_mfma_kernel<X3, OUT_BF16>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
        attr_set = true;
    }
    const int ntx = (X + TX - 1) / TX, nty = (Y + TY - 1) / TY, ntz = (Z + TZ - 1) / TZ;
    const int64_t nt = (int64_t)B * ntx * nty * ntz;
    if (nt > 0x7fffffff) return MMR_EINVAL;
    const int grid = nt < 2048 ? (int)nt : 2048;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(M2_THREADS), lds, st, src, trg, w, bias, out, pool, B, X, Y, Z, Cout,
                       leaky, alpha, ntx, nty, ntz, (int)nt);
    return check_launch();
}

extern "C" int mmr_conv3d_k3_cin2_fwd(const float* src, const float* trg, const float* w_keras, const float* bias,
                                      void* out, void* pool_out, int B, int X, int Y, int Z, int Cout, int leaky,
                                      float alpha, int out_dtype, void* stream)
{
    if (!src || !trg || !w_keras || !out || B < 1 || X < 1 || Y < 1 || Z < 1 || Cout < 1) return MMR_EINVAL;
    if (pool_out && (X < 2 || Y < 2 || Z < 2)) return MMR_EINVAL;
    if (leaky && !(alpha >= 0.f && alpha <= 1.f)) return MMR_EINVAL;   // the matrix-core kernels apply max(v, alpha v)
    hipStream_t st = as_stream(stream);
    // out_dtype: BF16 -> bf16 MFMA, bf16 out; F32X3 -> bf16 hi/lo split MFMA, fp32 out; F32 -> exact fp32 VALU kernel
    // pool_out (fused MaxPooling3D(2) of the activated output, same element type): the two matrix-core kernels only
    if (Cout % 64 == 0 && Cout <= 512 && out_dtype == MMR_DT_BF16)   // a 128-B output line = 64 bf16 couts
        return launch_cin2_mfma<false, true>(src, trg, w_keras, bias, out, pool_out, B, X, Y, Z, Cout, leaky, alpha, st);
    if (Cout % 32 == 0 && Cout <= 320 && out_dtype == MMR_DT_F32X3)
        return launch_cin2_mfma<true, false>(src, trg, w_keras, bias, out, pool_out, B, X, Y, Z, Cout, leaky, alpha, st);
    if (pool_out) return MMR_EUNSUPPORTED;   // exact-fp32 VALU kernel: use mmr_maxpool3d2_fwd
    if (out_dtype == MMR_DT_F32X3) out_dtype = MMR_DT_F32;
    if (!((Cout <= 256 && 256 % Cout == 0) || Cout % 256 == 0)) return MMR_EINVAL;
    const int ntx = (X + F_TX - 1) / F_TX, nty = (Y + F_TY - 1) / F_TY, ntz = (Z + F_TZ - 1) / F_TZ;
    const int64_t nblk = (int64_t)B * ntx * nty * ntz;
    if (nblk > 0x7fffffff) return MMR_EINVAL;
    if (out_dtype == MMR_DT_BF16)
        hipLaunchKernelGGL(conv3d_cin2_kernel<MMR_DT_BF16>, dim3((unsigned)nblk), dim3(256), 0, st, src,
                           trg, w_keras, bias, out, B, X, Y, Z, Cout, leaky, alpha, ntx, nty, ntz);
    else if (out_dtype == MMR_DT_F32)
        hipLaunchKernelGGL(conv3d_cin2_kernel<MMR_DT_F32>, dim3((unsigned)nblk), dim3(256), 0, st, src,
                           trg, w_keras, bias, out, B, X, Y, Z, Cout, leaky, alpha, ntx, nty, ntz);
    else
        return MMR_EINVAL;
    return check_launch();
}

// Flow head: Conv3D(3, 3, 'same', no activation), in [B,X,Y,Z,Cin] (bf16, or fp32 with MMR_DT_F32X3) -> out fp32.
extern "C" int mmr_conv3d_k3_cout3_fwd(const void* in, const float* w_keras, const float* bias, float* out, int B,
                                       int X, int Y, int Z, int Cin, int dtype, void* stream)
{
    if (!in || !w_keras || !out || B < 1 || X < 1 || Y < 1 || Z < 1 || Cin < 32 || Cin % 32) return MMR_EINVAL;
    if (dtype != MMR_DT_BF16 && dtype != MMR_DT_F32X3) return MMR_EUNSUPPORTED;
    const int npl = dtype == MMR_DT_F32X3 ? 2 : 1;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipSuccess;
        const void* ks[11] = {reinterpret_cast<const void*>(flow_head_march_kernel<false, 8, 16>),
                             reinterpret_cast<const void*>(flow_head_kernel<false>), reinterpret_cast<const void*>(flow_head_kernel<true>),
                             reinterpret_cast<const void*>(flow_head_march_kernel<true, 4>),
                             reinterpret_cast<const void*>(flow_head_march_kernel<true, 2>),
                             reinterpret_cast<const void*>(flow_head_march_kernel<true, 1>),
                             reinterpret_cast<const void*>(flow_head_march_kernel<false>),
                             reinterpret_cast<const void*>(flow_head_march_kernel<true>),
                             reinterpret_cast<const void*>(flow_head_march_kernel<false, 8>),
                             reinterpret_cast<const void*>(flow_head_march_kernel<false, 4>),
                             reinterpret_cast<const void*>(flow_head_march_kernel<false, 2>)};
        for (int i = 0; i < 11 && e == hipSuccess; ++i)
            e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_hip_error(e); return MMR_EHIP; }
        attr_set = true;
    }
    const int lds_w = npl * (Cin / 8) * 96 * 16;
    // bf16 with 256 input channels (BASELINE configs[1]): 16 x 16-row patches, eight waves, one P plane (see the kernel's HY)
    if (dtype == MMR_DT_BF16 && Cin == 256 && Y >= 28 && lds_w + 16 * MH_HZ * 81 * 4 <= 160 * 1024) {
        constexpr int HY16 = 16;
        const int nty = (Y + (HY16 - 2) - 1) / (HY16 - 2), ntz = (Z + MH_TZ - 1) / MH_TZ;
        const int64_t tyz = (int64_t)B * nty * ntz;
        const int gmax = 256;
        int nseg = 1;
        {
            int64_t best = -1;
            const int smax = (X + 3) / 4;
            for (int c = 1; c <= smax && c <= 64; ++c) {
                const int sl = (X + c - 1) / c;
                const int ns = (X + sl - 1) / sl;
                const int64_t per = (tyz * ns + gmax - 1) / gmax;
                const int64_t cost = per * (sl + 2);
                if (best < 0 || cost < best) { best = cost; nseg = ns; }
            }
        }
        const int seglen = (X + nseg - 1) / nseg;
        nseg = (X + seglen - 1) / seglen;
        const int64_t nt = tyz * nseg;
        if (nt > 0x7fffffff) return MMR_EINVAL;
        const int grid = nt < gmax ? (int)nt : gmax;
        hipLaunchKernelGGL((flow_head_march_kernel<false, 8, HY16>), dim3(grid), dim3(HY16 * 32), lds_w + HY16 * MH_HZ * 81 * 4,
                           as_stream(stream), (const char*)in, w_keras, bias, out, B, X, Y, Z, Cin, nseg, seglen, nty, ntz, (int)nt, 1);
        return check_launch();
    }
    // double-buffered P plane when it fits, else a single buffer (one more barrier per plane): fp32x3 at Cin = 256
    int pbufs = (lds_w + 2 * MH_ROWS * 81 * 4 <= 160 * 1024) ? 2 : 1;
    // narrow inputs: two workgroups per CU with a single P buffer each (one more barrier per plane, but a second workgroup
    // to cover it: 0.52 -> 0.40 ms for fp32x3 at 160^3 x 64)
    const bool two_wg = 2 * (lds_w + MH_ROWS * 81 * 4) <= 160 * 1024;
    if (two_wg) pbufs = 1;
    const int lds_m = lds_w + pbufs * MH_ROWS * 81 * 4;
    if (lds_m <= 160 * 1024) {   // else: the 2x4x8-tile kernel below (weights + one P plane do not fit)
        const int nty = (Y + MH_TY - 1) / MH_TY, ntz = (Z + MH_TZ - 1) / MH_TZ;
        const int64_t tyz = (int64_t)B * nty * ntz;
        // x segments: the makespan of the persistent grid is (tiles of the busiest workgroup) x (planes per tile, + 2 of
        // halo); pick the segment count that minimises it (at C2: 2 segments = 3 x 82 plane steps against 5 x 56 for the
        // "about 1024 tiles" rule this replaces)
        const int gmax = two_wg ? 512 : 256;
        int nseg = 1;
        {
            int64_t best = -1;
            const int smax = (X + 3) / 4;
            for (int c = 1; c <= smax && c <= 64; ++c) {
                const int sl = (X + c - 1) / c;
                const int ns = (X + sl - 1) / sl;
                const int64_t per = (tyz * ns + gmax - 1) / gmax;
                const int64_t cost = per * (sl + 2);
                if (best < 0 || cost < best) { best = cost; nseg = ns; }
            }
        }
        const int seglen = (X + nseg - 1) / nseg;
        nseg = (X + seglen - 1) / seglen;
        const int64_t nt = tyz * nseg;
        if (nt > 0x7fffffff) return MMR_EINVAL;
        const int grid = nt < gmax ? (int)nt : gmax;
        if (dtype == MMR_DT_F32X3 && Cin == 128)
            hipLaunchKernelGGL((flow_head_march_kernel<true, 4>), dim3(grid), dim3(MH_THREADS), lds_m, as_stream(stream),
                               (const char*)in, w_keras, bias, out, B, X, Y, Z, Cin, nseg, seglen, nty, ntz, (int)nt, pbufs);
        else if (dtype == MMR_DT_F32X3 && Cin == 64)
            hipLaunchKernelGGL((flow_head_march_kernel<true, 2>), dim3(grid), dim3(MH_THREADS), lds_m, as_stream(stream),
                               (const char*)in, w_keras, bias, out, B, X, Y, Z, Cin, nseg, seglen, nty, ntz, (int)nt, pbufs);
        else if (dtype == MMR_DT_F32X3 && Cin == 32)
            hipLaunchKernelGGL((flow_head_march_kernel<true, 1>), dim3(grid), dim3(MH_THREADS), lds_m, as_stream(stream),
                               (const char*)in, w_keras, bias, out, B, X, Y, Z, Cin, nseg, seglen, nty, ntz, (int)nt, pbufs);
        else if (dtype == MMR_DT_F32X3)
            hipLaunchKernelGGL(flow_head_march_kernel<true>, dim3(grid), dim3(MH_THREADS), lds_m, as_stream(stream),
                               (const char*)in, w_keras, bias, out, B, X, Y, Z, Cin, nseg, seglen, nty, ntz, (int)nt, pbufs);
        else if (Cin == 256)
            hipLaunchKernelGGL((flow_head_march_kernel<false, 8>), dim3(grid), dim3(MH_THREADS), lds_m, as_stream(stream),
                               (const char*)in, w_keras, bias, out, B, X, Y, Z, Cin, nseg, seglen, nty, ntz, (int)nt, pbufs);
        else if (Cin == 128)
            hipLaunchKernelGGL((flow_head_march_kernel<false, 4>), dim3(grid), dim3(MH_THREADS), lds_m, as_stream(stream),
                               (const char*)in, w_keras, bias, out, B, X, Y, Z, Cin, nseg, seglen, nty, ntz, (int)nt, pbufs);
        else if (Cin == 64)
            hipLaunchKernelGGL((flow_head_march_kernel<false, 2>), dim3(grid), dim3(MH_THREADS), lds_m, as_stream(stream),
                               (const char*)in, w_keras, bias, out, B, X, Y, Z, Cin, nseg, seglen, nty, ntz, (int)nt, pbufs);
        else
            hipLaunchKernelGGL(flow_head_march_kernel<false>, dim3(grid), dim3(MH_THREADS), lds_m, as_stream(stream),
                               (const char*)in, w_keras, bias, out, B, X, Y, Z, Cin, nseg, seglen, nty, ntz, (int)nt, pbufs);
        return check_launch();
    }
    const int lds = npl * (Cin / 8) * 96 * 16 + FH_ROWS * 81 * 4;
    if (lds > 160 * 1024) return MMR_EUNSUPPORTED;
    const int ntx = (X + FH_TX - 1) / FH_TX, nty = (Y + FH_TY - 1) / FH_TY, ntz = (Z + FH_TZ - 1) / FH_TZ;
    const int64_t nt = (int64_t)B * ntx * nty * ntz;
    if (nt > 0x7fffffff) return MMR_EINVAL;
    const int grid = nt < 1024 ? (int)nt : 1024;
    if (dtype == MMR_DT_F32X3)
        hipLaunchKernelGGL(flow_head_kernel<true>, dim3(grid), dim3(FH_THREADS), lds, as_stream(stream), (const char*)in,
                           w_keras, bias, out, B, X, Y, Z, Cin, ntx, nty, ntz, (int)nt);
    else
        hipLaunchKernelGGL(flow_head_kernel<false>, dim3(grid), dim3(FH_THREADS), lds, as_stream(stream), (const char*)in,
                           w_keras, bias, out, B, X, Y, Z, Cin, ntx, nty, ntz, (int)nt);
    return check_launch();
}

extern "C" int mmr_maxpool3d2_fwd(const void* in, void* out, int B, int X, int Y, int Z, int C, int dtype, void* stream)
{
    if (!in || !out || B < 1 || X < 2 || Y < 2 || Z < 2 || C < 1) return MMR_EINVAL;
    const int epc = (dtype == MMR_DT_BF16) ? 8 : 4;
    if (dtype != MMR_DT_BF16 && dtype != MMR_DT_F32) return MMR_EINVAL;
    if (C % epc) return MMR_EINVAL;
    const int64_t total = (int64_t)B * (X / 2) * (Y / 2) * (Z / 2) * (C / epc);
    if (dtype == MMR_DT_BF16)
        hipLaunchKernelGGL(maxpool_kernel<MMR_DT_BF16>, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream),
                           (const char*)in, (char*)out, B, X, Y, Z, C);
    else
        hipLaunchKernelGGL(maxpool_kernel<MMR_DT_F32>, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream),
                           (const char*)in, (char*)out, B, X, Y, Z, C);
    return check_launch();
}
