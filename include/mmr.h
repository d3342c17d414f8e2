/* mmr.h -- C-ABI of the MI355X-native SynthMorph/VoxelMorph hot path.
 *
 * The reference (ivadomed/multimodal-registration) has no FFI: its boundary is
 * the Python operator surface of voxelmorph/neurite.
 * Each entry point below names the reference call site whose arithmetic it
 * replaces; `multimodal-registration_amd/` binds them with ctypes and mirrors
 * the reference's operator names on top.
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative MMR_E* code otherwise; never throw,
 *     never allocate or free caller memory; outputs and scratch are
 *     caller-provided DEVICE pointers (gfx950), NDHWC contiguous;
 *   - `stream` is a hipStream_t passed as void*; work is only ordered by it;
 *   - stateless and re-entrant.
 */
#ifndef MMR_H
#define MMR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMR_OK 0
#define MMR_EINVAL (-1)     /* bad shape / null pointer / unsupported combination */
#define MMR_EHIP (-2)       /* HIP launch or runtime error */
#define MMR_EUNSUPPORTED (-3)

#define MMR_INTERP_LINEAR 0
#define MMR_INTERP_NEAREST 1

#define MMR_DT_F32 0
#define MMR_DT_BF16 1
#define MMR_DT_F32X3 2  /* fp32 tensors, bf16 hi/lo split inside the conv (3 bf16 MFMAs per product) */
#define MMR_DT_F32X1 3  /* fp32 tensors, products of the bf16 hi halves only (opt-in, backward pass) */

/* Upstream semantics that cannot be pinned here (voxelmorph / neurite are absent, SURVEY.md Appendix A "warning"
 * items A4, A6, A8) are selectable per call; 0 is the behaviour recalled for the commits the reference pins
 * (README.md:35-37) and the default of the Python layer (mmr.semantics). */
#define MMR_RESIZE_ALIGN_CORNERS 0 /* A4: sample grid linspace(0, old-1, new)  (neurite, late 2021) */
#define MMR_RESIZE_ARANGE_OVER_F 1 /* A4: sample grid arange(new) / zoom, clamped at the edge (older neurite) */
#define MMR_DICE_DIVIDE_NO_NAN 0   /* A6: tf.math.divide_no_nan(top, bottom) */
#define MMR_DICE_MAX_EPS 1         /* A6: top / max(bottom, 1e-5)            (older voxelmorph) */
#define MMR_NCC_CLASSIC 0          /* A8: cc = cross^2 / (I_var * J_var + eps) */
#define MMR_NCC_CLAMPED 1          /* A8: cross, I_var, J_var clamped to >= eps; cc = (cross / I_var) * (cross / J_var) */

int mmr_version(void);
const char* mmr_error_string(int code);
/* last HIP error text recorded by a failing call on this thread */
const char* mmr_last_hip_error(void);

/* ---- spatial transform ------------------------------------------------- *
 * vxm.layers.SpatialTransformer / vxm.utils.transform / ne.utils.interpn
 * (train_synthmorph.py:67,298; 3d_reg.py:331-334,377-380; inside VxmDense).
 * out[b,x,c] = vol[b, x + flow[b,x,:], c]; voxel units, 'ij'; clamp-to-edge;
 * if has_fill, positions with a coordinate <0 or >max get `fill`.
 * flow: [B,X,Y,Z,3], or [B,X,Y,Z,C,3] when channelwise != 0.             */
int mmr_warp3d_f32(const float* vol, const float* flow, float* out,
                   int B, int X, int Y, int Z, int C,
                   int interp, int has_fill, float fill, int channelwise, void* stream);
/* nearest-neighbour label resampling, bit-exact (labels_to_image warp). */
int mmr_warp3d_nearest_u8(const uint8_t* vol, const float* flow, uint8_t* out,
                          int B, int X, int Y, int Z, int C,
                          int has_fill, uint8_t fill, void* stream);

/* ne.utils.resize + vxm RescaleTransform (inside VxmDense; 3d_reg.py:394):
 * trilinear from (X,Y,Z) to (Xo,Yo,Zo), values times `mul`.
 * pre_scale!=0 multiplies before interpolating (factor>1 branch).
 * grid_mode: MMR_RESIZE_* (A4). zoom: the zoom factor f of MMR_RESIZE_ARANGE_OVER_F (0 = derive new/old
 * per axis); ignored for MMR_RESIZE_ALIGN_CORNERS.                          */
int mmr_resize_trilinear_f32(const float* in, float* out, int B, int X, int Y, int Z, int C,
                             int Xo, int Yo, int Zo, float mul, int pre_scale, int grid_mode, float zoom,
                             void* stream);

/* vxm.utils.compose([A,B]) (bids_two_steps_registration.py:324):
 * out = B + A o (id + B). Also one scaling-and-squaring step when a == b. */
int mmr_compose_f32(const float* a, const float* b, float* out,
                    int B, int X, int Y, int Z, void* stream);

/* vxm.layers.VecInt('ss', int_steps) (config.json:41): out = integrate(vel).
 * `tmp` is scratch of the same size as vel; vel is not modified.           */
int mmr_vecint_f32(const float* vel, float* out, float* tmp,
                   int B, int X, int Y, int Z, int nsteps, void* stream);

/* ---- U-Net ------------------------------------------------------------- *
 * Conv3D(nf,3,'same')+bias(+LeakyReLU) of VxmDense's U-Net
 * (train_synthmorph.py:296; 3d_reg.py:305).  Implicit GEMM on MFMA.
 * Weights are first re-packed from the Keras layout [3][3][3][Cin][Cout]
 * (fp32, device) into the kernel's tap-major MFMA operand image.           */
int64_t mmr_conv3d_k3_packed_bytes(int Cin, int Cout, int dtype);
int mmr_conv3d_k3_pack(const float* w_keras, void* w_packed, int Cin, int Cout, int dtype,
                       int transpose_flip, void* stream);
/* in0: [B,X0,Y0,Z0,C0] (if up0: at half resolution, nearest-upsampled on the
 * fly = UpSampling3D(2)); in1: optional skip [B,X,Y,Z,C1] concatenated AFTER
 * in0's channels (= concatenate([upsampled, skip])).  dtype selects element
 * type of in0/in1/w_packed/out (bf16 in, fp32 accumulate, or exact fp32).
 * out_f32 != 0 stores fp32 output even for the bf16 path.
 * pool_out: must be NULL here (MMR_EUNSUPPORTED otherwise).  The pooling pass that costs time is the one after the
 * FIRST layer, and that one is fused: mmr_conv3d_k3_cin2_fwd.  After the deeper encoder layers the tensors are
 * 8x .. 512x smaller (0.1 ms in total at 160x160x192 x 256); use mmr_maxpool3d2_fwd.                          */
int mmr_conv3d_k3_fwd(const void* in0, int C0, int up0, const void* in1, int C1,
                      const void* w_packed, const float* bias, void* out, void* pool_out,
                      int B, int X, int Y, int Z, int Cout,
                      int leaky, float alpha, int dtype, int out_f32, void* stream);
/* Same, with an optional scratch buffer of mmr_conv3d_k3_ksplit_ws_bytes(): launches with too few workgroups to
 * fill the chip (deep U-Net levels, small volumes) then split the K walk over several workgroups per tile and add
 * the partial tiles in a fixed order (bitwise reproducible).  A launch whose tile count leaves a partial round of
 * workgroups (one per CU: e.g. 300 tiles on 256 CUs) runs only its LAST nblk % CUs tiles that way, in a second launch
 * (from 4 x 27 tap steps per tile on).  The query returns 0 when neither applies.  ws == NULL behaves like
 * mmr_conv3d_k3_fwd. */
int64_t mmr_conv3d_k3_ksplit_ws_bytes(int B, int X, int Y, int Z, int Cin, int Cout, int dtype);
int mmr_conv3d_k3_fwd_ws(const void* in0, int C0, int up0, const void* in1, int C1,
                         const void* w_packed, const float* bias, void* out, void* pool_out,
                         int B, int X, int Y, int Z, int Cout,
                         int leaky, float alpha, int dtype, int out_f32, void* ws, void* stream);
/* Folded upsampling -- the decoder layers of VxmDense convolve concatenate([UpSampling3D(2)(x), skip])
 * (3d_reg.py:297-305 builds four of them).  For the upsampled channels the 27 taps at a
 * full-resolution voxel 2i+p touch only a 2x2x2 block of x, so that half of the layer is 8 convolutions (one per
 * parity class p) with 8 taps each on the LOW-resolution grid with pre-summed weights: 8/27 of the multiply-adds.
 *   mmr_conv3d_k3_upfold_pack: w_up_keras [27][C0][Cout] fp32 (the first C0 input channels of the layer's Keras
 *       kernel, contiguous) -> folded MFMA operand image of mmr_conv3d_k3_upfold_packed_bytes() bytes;
 *   mmr_conv3d_k3_upfold_fwd:  in_low [B,X2,Y2,Z2,C0] (bf16 for MMR_DT_BF16, fp32 for MMR_DT_F32X3) ->
 *       partial [B,2*X2,2*Y2,2*Z2,Cout] (raw sums, no bias; every element written) in fp32, or -- partial_half != 0,
 *       MMR_DT_BF16 only -- in IEEE half saturated to +-65504 (its 2^-12 rounding is below the layer's bf16 output
 *       rounding; halves the round trip of the partial tensor through HBM);
 *   mmr_conv3d_k3_fwd_init:    out = act(cinit + conv3x3x3(in) + bias) over the skip channels, cinit = that partial
 *       (cinit_half says which of the two formats it holds); ws: NULL, or mmr_conv3d_k3_ksplit_ws_bytes(B, X, Y, Z, Cin,
 *       Cout, dtype) bytes of scratch (see there).
 * The pair equals mmr_conv3d_k3_fwd(in_low, C0, up0 = 1, skip, C1, ...) up to the summation order of the weights.
 * bf16 / fp32x3 only, C0 a multiple of the channel slice (64 / 32), Cout a multiple of 64; else MMR_EUNSUPPORTED /
 * MMR_EINVAL. */
int64_t mmr_conv3d_k3_upfold_packed_bytes(int C0, int Cout, int dtype);
int mmr_conv3d_k3_upfold_pack(const float* w_up_keras, void* w_packed, int C0, int Cout, int dtype, void* stream);
int mmr_conv3d_k3_upfold_fwd(const void* in_low, int C0, const void* w_packed, void* partial, int partial_half,
                             int B, int X2, int Y2, int Z2, int Cout, int dtype, void* stream);
int mmr_conv3d_k3_fwd_init(const void* in, int Cin, const void* w_packed, const float* bias, const void* cinit,
                           int cinit_half, void* out, int B, int X, int Y, int Z, int Cout,
                           int leaky, float alpha, int dtype, int out_f32, void* ws, void* stream);
/* Data gradient of the folded half w.r.t. the low-resolution input (training; fp32x3 / x1): replaces the upsampled-channel
 * half of mmr_conv3d_k3_dgrad_split plus the 2x2x2 pooling of mmr_upcat_bwd_masked_f32 -- 64 tap-steps per low-res voxel
 * instead of 216, no full-resolution intermediate.  dz [B,2*X2,2*Y2,2*Z2,Cz] fp32 -> out [B,X2,Y2,Z2,C0] fp32;
 * w_up_keras [27][C0][Cz] = the first C0 input channels of the layer's FORWARD Keras kernel.  ymask != NULL: out is
 * multiplied by LeakyReLU'(ymask) and dbias (+)= its column sums (ws of mmr_conv3d_k3_dgrad_upfold_ws_bytes). */
int64_t mmr_conv3d_k3_dgrad_upfold_packed_bytes(int Cz, int C0, int dtype);
int mmr_conv3d_k3_dgrad_upfold_pack(const float* w_up_keras, void* w_packed, int C0, int Cz, int dtype, void* stream);
int64_t mmr_conv3d_k3_dgrad_upfold_ws_bytes(int B, int X2, int Y2, int Z2, int C0);
/* Every weight image of a training step in ONE launch (after the optimizer step all of them are stale at once:
 * train_synthmorph.py:296-308 runs forward + backward on the weights Adam just wrote).  A job names a Keras kernel
 * w [27][rows_total][cols] fp32 (the layer's whole kernel, in place) and the input-channel rows row_off .. row_off + rows
 * of it, so the channel slices of a concat layer need no contiguous copy:
 *   MMR_PACK_FWD     = mmr_conv3d_k3_pack(w_slice, out, Cin = rows, Cout = cols, transpose_flip = 0)
 *   MMR_PACK_DGRAD   = mmr_conv3d_k3_pack(w_slice, out, Cin = cols, Cout = rows, transpose_flip = 1)
 *   MMR_PACK_UPFOLD  = mmr_conv3d_k3_upfold_pack(w_slice, out, C0 = rows, Cout = cols)
 *   MMR_PACK_DGFOLD  = mmr_conv3d_k3_dgrad_upfold_pack(w_slice, out, C0 = rows, Cz = cols)
 * bit for bit; out holds mmr_conv3d_k3_pack_job_bytes(kind, rows, cols, dtype) bytes.  `jobs` is a host array. */
#define MMR_PACK_FWD 0
#define MMR_PACK_DGRAD 1
#define MMR_PACK_UPFOLD 2
#define MMR_PACK_DGFOLD 3
typedef struct MmrPackJob {
    const float* w;
    void* out;
    int32_t kind, rows_total, row_off, rows, cols, reserved;
} MmrPackJob;
int64_t mmr_conv3d_k3_pack_job_bytes(int kind, int rows, int cols, int dtype);
int mmr_conv3d_k3_pack_batch(const MmrPackJob* jobs, int njobs, int dtype, void* stream);
int mmr_conv3d_k3_dgrad_upfold(const void* dz, int Cz, const void* w_packed, float* out, int B, int X2, int Y2, int Z2,
                               int C0, const float* ymask, float alpha, float* dbias, void* ws, int accumulate,
                               int dtype, void* stream);
/* Weight gradient of the same layer, folded: dw [27][C0 + C1][Cout] (+)= dL/dW given dz [B,2*X2,2*Y2,2*Z2,Cout].  Rows [0, C0)
 * (the upsampled channels) come from the low-resolution x_low [B,X2,Y2,Z2,C0]: per parity class p the 8 correlations
 * dWf[p][s] = sum_i x_low[i - 1 + p + s] (x) dz[2 i + p] on the low-resolution grid, each added to the original taps its fold
 * covers (64 class-tap products over N / 8 voxels instead of 27 over N); rows [C0, C0 + C1) from skip [B,2*X2,..,C1] through the
 * ordinary kernel.  x3mode 1 = fp32x3 products, 2 = bf16 hi products only.  Cout a multiple of 64, C0 / C1 of 32, every tensor
 * under 3.75 GB; else MMR_EINVAL / MMR_EUNSUPPORTED (use mmr_conv3d_k3_wgrad_f32x3 with up0 = 1). */
int64_t mmr_conv3d_k3_wgrad_upfold_ws_bytes(int B, int X2, int Y2, int Z2, int C0, int C1, int Cout);
int mmr_conv3d_k3_wgrad_upfold(const float* x_low, int C0, const float* skip, int C1, const float* dz, float* dw, void* ws,
                               int B, int X2, int Y2, int Z2, int Cout, int accumulate, int x3mode, void* stream);
/* First layer: concatenate([moving, fixed]) (2 x 1 channel, fp32) -> Cout.  pool_out (optional, same element type as
 * out, [B,X/2,Y/2,Z/2,Cout]): MaxPooling3D(2) of the activated output from the same kernel (bf16 and fp32x3 kernels;
 * MMR_EUNSUPPORTED with the exact-fp32 kernel).  Saves the 2.5 GB read of a separate pooling pass at C2.
 * leaky != 0: LeakyReLU slope alpha in [0, 1] (the kernels apply max(v, alpha v)); MMR_EINVAL otherwise.          */
int mmr_conv3d_k3_cin2_fwd(const float* src, const float* trg, const float* w_keras, const float* bias,
                           void* out, void* pool_out, int B, int X, int Y, int Z, int Cout,
                           int leaky, float alpha, int out_dtype, void* stream);
/* Flow head Conv3D(3, 3, 'same') without activation: taps folded into the GEMM N axis (81 of 96 columns).
 * in: bf16 (MMR_DT_BF16) or fp32 (MMR_DT_F32X3); out fp32 [B,X,Y,Z,3]. */
int mmr_conv3d_k3_cout3_fwd(const void* in, const float* w_keras, const float* bias, float* out,
                            int B, int X, int Y, int Z, int Cin, int dtype, void* stream);
int mmr_maxpool3d2_fwd(const void* in, void* out, int B, int X, int Y, int Z, int C, int dtype, void* stream);

/* ---- losses ------------------------------------------------------------ *
 * vxm.losses.Dice().loss (train_synthmorph.py:306). ws: >= mmr_dice_ws_bytes.
 * loss_out[0] = -mean_{b,l} ratio(2 sum(t p), sum(t + p)); ratio per dice_mode (MMR_DICE_*, A6). */
int64_t mmr_dice_ws_bytes(int B, int64_t nvox, int L);
int mmr_dice_fwd_f32(const float* y_true, const float* y_pred, float* loss_out, float* top_bot,
                     void* ws, int B, int64_t nvox, int L, int dice_mode, void* stream);
/* losses.dice_loss_zeropad on dense maps (losses.py:11-69 as its docstring intends; the reference function always
 * raises): voxels whose channel 0 is >= 1 in either map are zeroed, loss = -mean over labels 1..L-1 of batch item 0. */
int mmr_dice_zeropad_fwd_f32(const float* y_true, const float* y_pred, float* loss_out, float* top_bot,
                             void* ws, int B, int64_t nvox, int L, int dice_mode, void* stream);
/* d loss / d y_pred from the forward's (top, bot) sums [B,L,2]; dpred (+)= scale * gradient. */
int mmr_dice_bwd_f32(const float* y_true, const float* top_bot, float* dpred, int B, int64_t nvox, int L,
                     float scale, int accumulate, int dice_mode, void* stream);
/* vxm.losses.Grad('l2', loss_mult).loss(None, flow) (train_synthmorph.py:307) -> out[B]. */
int64_t mmr_grad_l2_ws_bytes(int B, int X, int Y, int Z, int C);
int mmr_grad_l2_fwd_f32(const float* flow, float* out, void* ws,
                        int B, int X, int Y, int Z, int C, float loss_mult, void* stream);
/* vxm.losses.NCC(win) (BASELINE.json config 5; no reference call site) -> out[B]; ncc_form: MMR_NCC_* (A8). */
int64_t mmr_ncc_ws_bytes(int B, int X, int Y, int Z);
int mmr_ncc_fwd_f32(const float* I, const float* J, float* out, void* ws,
                    int B, int X, int Y, int Z, int win, float eps, int ncc_form, void* stream);
/* bending energy of a displacement field (config 5) -> out[B]. */
int64_t mmr_bending_ws_bytes(int B, int X, int Y, int Z);
int mmr_bending_fwd_f32(const float* flow, float* out, void* ws,
                        int B, int X, int Y, int Z, void* stream);
/* The two losses with the final reduction INSIDE the kernel (no second launch): out[b] = (accumulate ? out[b] : 0) + scale * loss_b,
 * so that `NCC + lambda * bending` lands in one tensor from two launches.  The workgroup that finishes last adds the per-workgroup
 * partials in index order (bitwise reproducible, the same sum as the two-launch entry points).  `ticket`: one 32-bit word of
 * DEVICE memory per concurrently running call; it must be ZERO when the call is issued and is zero again when its kernels have
 * finished (calls ordered on one stream may share it). */
int mmr_ncc_fwd_ticket_f32(const float* I, const float* J, float* out, void* ws, unsigned* ticket,
                           int B, int X, int Y, int Z, int win, float eps, int ncc_form, float scale, int accumulate,
                           void* stream);
int mmr_bending_fwd_ticket_f32(const float* flow, float* out, void* ws, unsigned* ticket,
                               int B, int X, int Y, int Z, float scale, int accumulate, void* stream);
/* Backward of the two losses (SURVEY 8b families ncc_bwd / bending_bwd): gradients of out[b] scaled by gout[b]
 * (gout == NULL: 1).  NCC: dI and/or dJ [B,X,Y,Z] (either may be NULL); workspace mmr_ncc_bwd_ws_bytes (five volumes for the
 * two-launch form, Z % 4 == 0 and Z <= 256: coefficient pass + one 9^3 box filter; 19 volumes for the separable form otherwise). */
int64_t mmr_ncc_bwd_ws_bytes(int B, int X, int Y, int Z);
int mmr_ncc_bwd_f32(const float* I, const float* J, const float* gout, float* dI, float* dJ, void* ws,
                    int B, int X, int Y, int Z, int win, float eps, int ncc_form, void* stream);
int mmr_bending_bwd_f32(const float* flow, const float* gout, float* dflow, int B, int X, int Y, int Z,
                        int accumulate, void* stream);

/* ---- SynthMorph generator (ne.models.labels_to_image / ne.utils.augment.draw_perlin,
 * train_synthmorph.py:57-64,258-268,288-291; stages SURVEY.md Appendix A9/A10) ---- */
/* Counter-based Philox4x32-10: out[i] depends only on (seed, stream_id, i). */
int mmr_philox_normal_f32(float* out, int64_t n, uint64_t seed, uint32_t stream_id, float mean, float std, void* stream);
int mmr_philox_uniform_f32(float* out, int64_t n, uint64_t seed, uint32_t stream_id, float lo, float hi, void* stream);
/* label look-up (in_label_list -> 0..L-1), lut256 on device */
int mmr_lut_u8(const uint8_t* in, uint8_t* out, const uint8_t* lut256, int64_t n, void* stream);
/* image = mean[b,label] + std[b,label] * N(0,1); `noise` (optional, [B,nvox]) overrides the Philox draw */
int mmr_gmm_sample_f32(const uint8_t* labels, const float* means, const float* stds, const float* noise, float* out,
                       int B, int64_t nvox, int L, uint64_t seed, uint32_t stream_id, void* stream);
/* one axis of the separable Gaussian blur, per-item kernel [B][W], 'SAME' zero padding */
int mmr_blur_axis_f32(const float* in, float* out, const float* kern, int B, int X, int Y, int Z, int axis, int W,
                      void* stream);
/* x <- ((clip(x*exp(bias), lo, hi) - min_b)/(max_b - min_b)) ** exp(gamma[b]), in place; bias, gamma optional */
int64_t mmr_intensity_ws_bytes(int B);
int mmr_bias_clip_norm_gamma_f32(float* x, const float* bias, const float* gamma, void* ws, int B, int64_t nvox,
                                 float lo, float hi, void* stream);
int mmr_onehot_f32(const uint8_t* labels, float* out, int64_t n, int L, void* stream);
/* tf.argmax(axis=-1) -> uint8 (generate_label_maps, train_synthmorph.py:68-69) */
int mmr_argmax_u8(const float* x, uint8_t* out, int64_t n, int C, void* stream);
int mmr_axpy_f32(float* y, const float* x, float a, int64_t n, void* stream);

/* ---- training: backward kernels + Adam (train_synthmorph.py:296-308,335-344), all fp32 ---- */
/* Dice(one_hot(lab2), SpatialTransformer('linear')(one_hot(lab1), flow)) computed from the uint8 label
 * volumes (never materialising the L-channel tensors): loss[0], top_bot[B][L][2] = (2 sum tp, sum t+p). */
int64_t mmr_dice_labels_ws_bytes(int B, int64_t nvox, int L);
int mmr_dice_labels_fwd(const uint8_t* lab1, const uint8_t* lab2, const float* flow, float* loss, float* top_bot,
                        void* ws, int B, int X, int Y, int Z, int L, int dice_mode, void* stream);
/* dflow (+)= scale * d loss / d flow */
int mmr_dice_labels_bwd(const uint8_t* lab1, const uint8_t* lab2, const float* flow, const float* top_bot,
                        float* dflow, int B, int X, int Y, int Z, int L, float scale, int accumulate, int dice_mode,
                        void* stream);
/* losses.dice_loss_zeropad (losses.py:11-69, as its docstring intends; the reference function itself always
 * raises): voxels whose label-0 channel is >= 1 in either map are masked, labels 1..L-1 of batch item 0. */
int mmr_dice_labels_zeropad_fwd(const uint8_t* lab1, const uint8_t* lab2, const float* flow, float* loss, float* top_bot,
                                void* ws, int B, int X, int Y, int Z, int L, int dice_mode, void* stream);
int mmr_dice_labels_zeropad_bwd(const uint8_t* lab1, const uint8_t* lab2, const float* flow, const float* top_bot,
                                float* dflow, int B, int X, int Y, int Z, int L, float scale, int accumulate,
                                int dice_mode, void* stream);
/* dflow (+)= scale * d/dflow sum_b Grad('l2', loss_mult)(flow)[b] */
int mmr_grad_l2_bwd_f32(const float* flow, float* dflow, int B, int X, int Y, int Z, int C, float loss_mult,
                        float scale, int accumulate, void* stream);
/* adjoint of mmr_resize_trilinear_f32 (din is overwritten); same grid_mode / zoom as the forward */
int mmr_resize_trilinear_bwd_f32(const float* dout, float* din, int B, int X, int Y, int Z, int C,
                                 int Xo, int Yo, int Zo, float mul, int grid_mode, float zoom, void* stream);
/* the same adjoint as three per-axis passes (the resize is separable) through a work space of
 * mmr_resize_trilinear_bwd_ws_bytes: 2 - 6 gathers per element and pass instead of their product; what the trainer uses */
int64_t mmr_resize_trilinear_bwd_ws_bytes(int B, int X, int Y, int Z, int C, int Xo, int Yo, int Zo);
int mmr_resize_trilinear_bwd_ws_f32(const float* dout, float* din, void* ws, int B, int X, int Y, int Z, int C,
                                    int Xo, int Yo, int Zo, float mul, int grid_mode, float zoom, void* stream);
/* adjoint of mmr_compose_f32: da, db overwritten (da == db allowed when a == b) */
int mmr_compose_bwd_f32(const float* a, const float* b, const float* dout, float* da, float* db,
                        int B, int X, int Y, int Z, void* stream);
/* VecInt forward keeping per-step inputs (steps: [nsteps-1][B,X,Y,Z,3]) and its adjoint */
int mmr_vecint_save_f32(const float* vel, float* steps, float* out, int B, int X, int Y, int Z, int nsteps, void* stream);
int mmr_vecint_bwd_f32(const float* vel, const float* steps, const float* dout, float* dvel, float* tmp,
                       int B, int X, int Y, int Z, int nsteps, void* stream);
/* SpatialTransformer('linear') gradients */
int mmr_warp3d_bwd_flow_f32(const float* vol, const float* flow, const float* dout, float* dflow,
                            int B, int X, int Y, int Z, int C, void* stream);
int mmr_warp3d_bwd_vol_f32(const float* flow, const float* dout, float* dvol, int B, int X, int Y, int Z, int C, void* stream);
/* dz = dy * LeakyReLU'(y) (y = activated output; in place allowed) and dbias (+)= sum_v dz */
int64_t mmr_leaky_bwd_ws_bytes(int64_t nvox, int C);
int mmr_leaky_bwd_bias_f32(const float* y, const float* dy, float* dz, float* dbias, void* ws, int64_t nvox, int C,
                           int leaky, float alpha, int accumulate, void* stream);
/* gradient of concatenate([UpSampling3D(2)(in0) | in0, in1]) split into its two sources */
int mmr_upcat_bwd_f32(const float* dcat, float* d_in0, float* d_in1, int B, int X, int Y, int Z, int C0, int C1,
                      int up0, int accumulate_in1, void* stream);
int mmr_maxpool3d2_bwd_f32(const float* x, const float* dpool, float* dx, int B, int X, int Y, int Z, int C,
                           int accumulate, void* stream);
/* conv weight gradients, Keras layout [27][Cin][Cout]; MFMA over voxels, ordered slab reduction */
int64_t mmr_conv3d_k3_wgrad_ws_bytes(int B, int X, int Y, int Z, int Cin, int Cout);
int mmr_conv3d_k3_wgrad_f32(const float* in0, int C0, int up0, const float* in1, int C1, const float* dz, float* dw,
                            void* ws, int B, int X, int Y, int Z, int Cout, int accumulate, void* stream);
/* same contract, bf16 hi/lo split products on the bf16 MFMA (pairs with MMR_DT_F32X3) */
int mmr_conv3d_k3_wgrad_f32x3(const float* in0, int C0, int up0, const float* in1, int C1, const float* dz, float* dw,
                              void* ws, int B, int X, int Y, int Z, int Cout, int accumulate, void* stream);
/* same, hi halves only (one bf16 MFMA per product, fp32 accumulate; pairs with MMR_DT_F32X1) */
int mmr_conv3d_k3_wgrad_f32x1(const float* in0, int C0, int up0, const float* in1, int C1, const float* dz, float* dw,
                              void* ws, int B, int X, int Y, int Z, int Cout, int accumulate, void* stream);
int64_t mmr_conv3d_k3_cin2_wgrad_ws_bytes(int Cout);
int mmr_conv3d_k3_cin2_wgrad_f32(const float* src, const float* trg, const float* dz, float* dw, void* ws,
                                 int B, int X, int Y, int Z, int Cout, int accumulate, void* stream);
/* same contract, bf16 hi/lo split products (pairs with MMR_DT_F32X3); widths that are not a multiple of 64 take the
 * exact path */
int mmr_conv3d_k3_cin2_wgrad_f32x3(const float* src, const float* trg, const float* dz, float* dw, void* ws,
                                   int B, int X, int Y, int Z, int Cout, int accumulate, void* stream);
/* flow head (Cout = 3) input gradient */
int mmr_conv3d_k3_cout3_dgrad_f32(const float* dy, const float* w_keras, float* dx, int B, int X, int Y, int Z, int Cin,
                                  void* stream);
/* same, bf16 hi/lo split products (pairs with MMR_DT_F32X3); Cin % 64 == 0 */
int mmr_conv3d_k3_cout3_dgrad_f32x3(const float* dy, const float* w_keras, float* dx, int B, int X, int Y, int Z, int Cin,
                                    void* stream);
/* Concat / upsample and max-pool backward with the same fusion: every gradient contribution to an activated tensor is
 * multiplied by LeakyReLU'(y) where it is produced and its column sums are added to that layer's bias gradient, so
 * no separate leaky-backward pass runs.  Channel counts must be multiples of 4 (else MMR_EUNSUPPORTED). */
int64_t mmr_upcat_bwd_masked_ws_bytes(int C0, int C1);
int mmr_upcat_bwd_masked_f32(const float* dcat, float* d_in0, float* d_in1, int B, int X, int Y, int Z, int C0, int C1,
                             int up0, int accumulate_in1, const float* y0, const float* y1, float alpha,
                             float* dbias0, int acc_b0, float* dbias1, int acc_b1, void* ws, void* stream);
int64_t mmr_maxpool3d2_bwd_masked_ws_bytes(int C);
int mmr_maxpool3d2_bwd_masked_f32(const float* x, const float* dpool, float* dx, int B, int X, int Y, int Z, int C,
                                  int accumulate, int masked, float alpha, float* dbias, int acc_b, void* ws,
                                  void* stream);
/* Data gradient of a layer whose input was concat([up2(in0) | in0, in1]), stored split: d0 = the C0 leading channels
 * (compact, to be pool-summed), d1 = the C1 skip channels times LeakyReLU'(y1) (y1 may be NULL) with dbias1 (+)= their
 * column sums -- the concatenated gradient is never materialised.  MMR_EUNSUPPORTED outside the 16x16x32 kernels. */
int64_t mmr_conv3d_k3_dgrad_split_ws_bytes(int B, int X, int Y, int Z, int C1);
int mmr_conv3d_k3_dgrad_split(const void* dz, int Cz, const void* w_packed, float* d0, float* d1, int B, int X, int Y, int Z,
                              int C0, int C1, const float* y1, float alpha, float* dbias1, void* ws, int accumulate,
                              int dtype, void* stream);
/* Same fusion for the flow head's data gradient (Cin % 64 == 0, else MMR_EUNSUPPORTED -> use the unfused pair). */
int64_t mmr_conv3d_k3_cout3_dgrad_masked_ws_bytes(int B, int X, int Y, int Z, int Cin);
int mmr_conv3d_k3_cout3_dgrad_masked_f32(const float* dy, const float* w_keras, float* dx, int B, int X, int Y, int Z,
                                         int Cin, const float* ymask, float alpha, float* dbias, void* ws,
                                         int accumulate, void* stream);
int mmr_conv3d_k3_cout3_dgrad_masked_f32x3(const float* dy, const float* w_keras, float* dx, int B, int X, int Y, int Z,
                                           int Cin, const float* ymask, float alpha, float* dbias, void* ws,
                                           int accumulate, void* stream);
/* Data gradient of a k3 conv fused with the LeakyReLU backward + bias gradient of the layer that produced the
 * conv's input (replaces mmr_leaky_bwd_bias_f32 after a dgrad; vxm Unet conv blocks):
 * out = conv(in0; w_packed = transposed/flipped weights) * (ymask < 0 ? alpha : 1); dbias (+)= sum_voxels out.
 * ymask = activated forward output of that layer, [B,X,Y,Z,Cout] fp32.  dtype: MMR_DT_F32 / F32X3 / F32X1. */
int64_t mmr_conv3d_k3_dgrad_masked_ws_bytes(int B, int X, int Y, int Z, int Cout);
int mmr_conv3d_k3_dgrad_masked(const void* in0, int C0, const void* w_packed, float* out, int B, int X, int Y, int Z,
                               int Cout, const float* ymask, float alpha, float* dbias, void* ws, int accumulate,
                               int dtype, void* stream);
/* the same with the gradient that reaches y (= ymask's tensor) through MaxPooling3D(2) folded into the epilogue:
 * out = (conv(in0) + route(dpool)) * LeakyReLU'(y), dbias (+)= column sums; dpool [B, X/2, Y/2, Z/2, Cout] goes to the first
 * maximum of every 2x2x2 window of y (window order x, y, z, like mmr_maxpool3d2_bwd_f32) -- the separate pooling-backward pass
 * over the full-resolution gradient disappears.  MMR_DT_F32X3 / F32X1, Cout % 64 == 0 and % 128 != 0, even X, Y, Z; else
 * MMR_EUNSUPPORTED.  Same work space as mmr_conv3d_k3_dgrad_masked. */
int mmr_conv3d_k3_dgrad_masked_pool(const void* in0, int C0, const void* w_packed, float* out, int B, int X, int Y, int Z,
                                    int Cout, const float* ymask, float alpha, float* dbias, void* ws, int accumulate,
                                    int dtype, const float* dpool, void* stream);
/* Keras Adam on one flat parameter buffer: g is multiplied by grad_scale first (1/world after a SUM all-reduce) */
int mmr_adam_step_f32(float* w, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, int64_t step, float grad_scale, void* stream);

/* ---- evaluation metrics of eval_reg_*.py, fp64 like the reference's NumPy ---- */
/* det(I + grad u) with 4th-order central differences (eval_reg_with_jacobian.py:62-78); ddf [X,Y,Z,3] */
int mmr_jacobian_det_f64(const double* ddf, double* det, int X, int Y, int Z, void* stream);
/* joint histogram with numpy.histogramdd semantics (eval_reg_with_mi.py:65-68); edges: nbins+1 each */
int mmr_joint_hist_f64(const double* a, const double* b, const double* edges_a, const double* edges_b,
                       unsigned long long* hist, int64_t n, int nbins, void* stream);
/* out6 = {sum m[f==1], sum m[f==0], #f==1, #f==0, sum m, n} (eval_reg_on_sc_seg.py:80-93) */
int64_t mmr_overlap_ws_bytes(void);
int mmr_overlap_sums_f64(const double* fixed, const double* moved, double* out6, void* ws, int64_t n, void* stream);

/* ---- host <-> device hand-over of model.predict([moving, fixed]) (3d_reg.py:310-314: nibabel get_fdata() float64
 * arrays in, NumPy arrays out).  The only entry points that touch HOST memory; they exist so that the host does no
 * arithmetic and no write into uncached memory on the way in:
 *   mmr_host_alloc      pinned, device-mapped staging memory; cached != 0 = CPU write-back pages (hipHostMallocNonCoherent)
 *   mmr_host_register   pins the caller's own buffer for the duration of a call and returns its device-side address
 *   mmr_cast_to_f32     kernel: n elements of MMR_HOST_* at src (pinned host OR device memory) -> fp32 in HBM, the
 *                       dtype conversion of Keras' predict and the H2D transfer in one pass over PCIe
 *   mmr_copy_to_host    kernel: device -> pinned host memory, bytes % 4 == 0                                          */
#define MMR_HOST_F64 0
#define MMR_HOST_F32 1
#define MMR_HOST_U8 2
#define MMR_HOST_I16 3
int mmr_host_alloc(void** out, int64_t bytes, int cached);
int mmr_host_free(void* p);
int mmr_host_register(void* p, int64_t bytes, void** dev_ptr);
int mmr_host_unregister(void* p);
int mmr_cast_to_f32(const void* src, float* dst, int64_t n, int src_dtype, void* stream);
int mmr_copy_to_host(const void* src_dev, void* dst_host, int64_t bytes, void* stream);
/* copy-engine transfer (no compute unit): pinned / registered host memory <-> device; kind 0 = host -> device, 1 = device -> host */
int mmr_memcpy_async(void* dst, const void* src, int64_t bytes, int kind, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMR_H */
