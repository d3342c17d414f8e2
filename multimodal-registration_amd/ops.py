"""Device-tensor wrappers over the C-ABI (include/mmr.h).

torch is used for device memory and streams only; every arithmetic operation
below is a call into libmmr_hip.so.  Shapes are channels-last.  All functions
raise on CPU tensors: there is no CPU path in the product.
"""
import torch

from . import _lib, semantics

F32, BF16, F32X3, F32X1 = 0, 1, 2, 3
PACK_FWD, PACK_DGRAD, PACK_UPFOLD, PACK_DGFOLD = _lib.PACK_FWD, _lib.PACK_DGRAD, _lib.PACK_UPFOLD, _lib.PACK_DGFOLD
LINEAR, NEAREST = 0, 1
_DT = {torch.float32: F32, torch.bfloat16: BF16}


def conv_mode(dtype, x3=False):
    """MFMA conv arithmetic: bf16 tensors -> BF16; fp32 tensors -> exact fp32 MFMA, or (x3) the bf16
    hi/lo split with three bf16 MFMAs per product (fp32-grade accuracy at ~5x the fp32-MFMA rate)."""
    if dtype == torch.bfloat16:
        return BF16
    if x3 == "hi":  # products of the bf16 hi halves only (opt-in for the backward pass)
        return F32X1
    return F32X3 if x3 else F32


def _stream():
    return torch.cuda.current_stream().cuda_stream


# bench.py sets PROFILE to a list to collect (kernel family, tag, ev0, ev1, algorithmic flops) per launch,
# timed with events on the stream the kernels are launched on (torch's current stream).
PROFILE = None


class _Timed:
    def __init__(self, family, tag, flops):
        self.rec = None
        if PROFILE is not None:
            self.rec = (family, tag, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), flops)

    def __enter__(self):
        if self.rec:
            self.rec[2].record()

    def __exit__(self, *a):
        if self.rec:
            self.rec[3].record()
            PROFILE.append(self.rec)


def _chk(t, dtype=None, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.MmrError(f"{name} must be a CUDA/HIP tensor (no CPU fallback in this package)")
    if dtype is not None and t.dtype != dtype:
        raise _lib.MmrError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.MmrError(f"{name} must be contiguous")
    return t


def interp_code(method):
    if method == "linear":
        return LINEAR
    if method == "nearest":
        return NEAREST
    raise ValueError(f"interp_method must be 'linear' or 'nearest', got {method!r}")


def warp3d(vol, flow, interp_method="linear", fill_value=None):
    """vol [B,X,Y,Z,C] f32, flow [B,X,Y,Z,3] (or [B,X,Y,Z,C,3] channel-wise) -> [B,X,Y,Z,C]."""
    _chk(vol, torch.float32, "vol")
    _chk(flow, torch.float32, "flow")
    B, X, Y, Z, C = vol.shape
    cw = 0
    if flow.dim() == 6:
        cw = 1
        if tuple(flow.shape) != (B, X, Y, Z, C, 3):
            raise _lib.MmrError(f"channel-wise flow shape {tuple(flow.shape)} != {(B, X, Y, Z, C, 3)}")
    elif tuple(flow.shape) != (B, X, Y, Z, 3):
        raise _lib.MmrError(f"flow shape {tuple(flow.shape)} does not match vol {tuple(vol.shape)}")
    out = torch.empty_like(vol)
    rc = _lib.load().mmr_warp3d_f32(vol.data_ptr(), flow.data_ptr(), out.data_ptr(), B, X, Y, Z, C,
                                    interp_code(interp_method), int(fill_value is not None),
                                    float(fill_value or 0.0), cw, _stream())
    _lib.check(rc, "mmr_warp3d_f32")
    return out


def warp3d_nearest_u8(vol, flow, fill_value=None):
    _chk(vol, torch.uint8, "vol")
    _chk(flow, torch.float32, "flow")
    B, X, Y, Z, C = vol.shape
    if tuple(flow.shape) != (B, X, Y, Z, 3):
        raise _lib.MmrError("flow shape mismatch")
    out = torch.empty_like(vol)
    rc = _lib.load().mmr_warp3d_nearest_u8(vol.data_ptr(), flow.data_ptr(), out.data_ptr(), B, X, Y, Z, C,
                                           int(fill_value is not None), int(fill_value or 0), _stream())
    _lib.check(rc, "mmr_warp3d_nearest_u8")
    return out


def resize_trilinear(x, out_shape, mul=1.0, pre_scale=False, grid=None, zoom=0.0):
    """Trilinear resize (ne.utils.resize) of [B,X,Y,Z,C] to out_shape=(Xo,Yo,Zo), values * mul.
    grid: 'align_corners' | 'arange_over_f' (None = mmr.semantics default); zoom: the zoom factor of the
    'arange_over_f' grid (0 = new/old per axis)."""
    _chk(x, torch.float32, "x")
    B, X, Y, Z, C = x.shape
    Xo, Yo, Zo = (int(s) for s in out_shape)
    out = torch.empty((B, Xo, Yo, Zo, C), dtype=torch.float32, device=x.device)
    rc = _lib.load().mmr_resize_trilinear_f32(x.data_ptr(), out.data_ptr(), B, X, Y, Z, C, Xo, Yo, Zo,
                                              float(mul), int(pre_scale), semantics.code("resize_grid", grid),
                                              float(zoom), _stream())
    _lib.check(rc, "mmr_resize_trilinear_f32")
    return out


def rescale_transform(trf, factor, grid=None):
    """vxm RescaleTransform / rescale_dense_transform on a batched field [B,X,Y,Z,3]."""
    B, X, Y, Z, _ = trf.shape
    new = (int(X * factor), int(Y * factor), int(Z * factor))
    return resize_trilinear(trf, new, mul=factor, pre_scale=factor >= 1, grid=grid, zoom=factor)


def compose(a, b):
    """out = b + a o (id + b), fields [B,X,Y,Z,3]."""
    _chk(a, torch.float32, "a")
    _chk(b, torch.float32, "b")
    if a.shape != b.shape or a.shape[-1] != 3:
        raise _lib.MmrError("compose needs two [B,X,Y,Z,3] fields of equal shape")
    B, X, Y, Z, _ = a.shape
    out = torch.empty_like(a)
    rc = _lib.load().mmr_compose_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), B, X, Y, Z, _stream())
    _lib.check(rc, "mmr_compose_f32")
    return out


def vecint(vel, nsteps):
    _chk(vel, torch.float32, "vel")
    if vel.shape[-1] != 3:
        raise _lib.MmrError("vecint needs [B,X,Y,Z,3]")
    B, X, Y, Z, _ = vel.shape
    out = torch.empty_like(vel)
    tmp = torch.empty_like(vel)
    rc = _lib.load().mmr_vecint_f32(vel.data_ptr(), out.data_ptr(), tmp.data_ptr(), B, X, Y, Z, int(nsteps), _stream())
    _lib.check(rc, "mmr_vecint_f32")
    return out


# ------------------------------- U-Net ---------------------------------- #
def conv_kc(dtype):
    return 64 if dtype == torch.bfloat16 else 32


def pack_conv_weights(w_keras, dtype, transpose_flip=False, x3=False):
    """w_keras [3,3,3,Cin,Cout] f32 (device) -> packed MFMA operand image (uint8 tensor)."""
    _chk(w_keras, torch.float32, "w_keras")
    cin, cout = int(w_keras.shape[3]), int(w_keras.shape[4])
    if transpose_flip:
        cin, cout = cout, cin
    lib = _lib.load()
    mode = conv_mode(dtype, x3)
    nbytes = lib.mmr_conv3d_k3_packed_bytes(cin, cout, mode)
    if nbytes < 0:
        raise _lib.MmrError(f"cannot pack conv weights Cin={cin} Cout={cout} for {dtype}")
    out = torch.empty(nbytes, dtype=torch.uint8, device=w_keras.device)
    rc = lib.mmr_conv3d_k3_pack(w_keras.data_ptr(), out.data_ptr(), cin, cout, mode, int(transpose_flip), _stream())
    _lib.check(rc, "mmr_conv3d_k3_pack")
    return out


def conv3d_k3(in0, w_packed, bias, cout, in1=None, up0=False, leaky=True, alpha=0.2, out_f32=False, x3=False):
    """Conv3D(cout,3,'same')(concat([up2(in0) if up0 else in0, in1])) + bias (+LeakyReLU)."""
    dtype = in0.dtype
    _chk(in0, dtype, "in0")
    B, X, Y, Z, C0 = in0.shape
    if up0:
        X, Y, Z = 2 * X, 2 * Y, 2 * Z
    C1 = 0
    if in1 is not None:
        _chk(in1, dtype, "in1")
        if tuple(in1.shape[:4]) != (B, X, Y, Z):
            raise _lib.MmrError(f"skip shape {tuple(in1.shape)} does not match {(B, X, Y, Z)}")
        C1 = in1.shape[4]
    odt = torch.float32 if (out_f32 or dtype == torch.float32) else torch.bfloat16
    out = torch.empty((B, X, Y, Z, cout), dtype=odt, device=in0.device)
    mode = conv_mode(dtype, x3)
    fam = f"conv3d_k3_mfma_{('f32', 'bf16', 'f32x3', 'f32x1')[mode]}_bn{256 if cout % 256 == 0 else 128 if cout % 128 == 0 else 64 if cout % 64 == 0 else 32}"
    lib = _lib.load()
    nws = lib.mmr_conv3d_k3_ksplit_ws_bytes(B, X, Y, Z, C0 + C1, int(cout), mode)  # > 0 only for launches that cannot fill the chip
    ws = _ws(nws, in0.device) if nws > 0 else None
    with _Timed(fam, (C0 + C1, int(cout), X, Y, Z), 2.0 * 27 * (C0 + C1) * cout * B * X * Y * Z):
        rc = lib.mmr_conv3d_k3_fwd_ws(
            in0.data_ptr(), C0, int(up0), in1.data_ptr() if in1 is not None else None, C1,
            w_packed.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), None,
            B, X, Y, Z, int(cout), int(leaky), float(alpha), mode, int(out_f32),
            ws.data_ptr() if ws is not None else None, _stream())
    _lib.check(rc, "mmr_conv3d_k3_fwd_ws")
    return out


def upfold_supported(C0, C1, cout, dtype, x3, B, X, Y, Z):
    """Folded upsampling (include/mmr.h, mmr_conv3d_k3_upfold_*) serves a decoder layer conv(concat([up2(x), skip])) when
    the element type is bf16 or fp32x3, the widths fit the MFMA slices and BOTH launches fill the chip on their own (the
    1/8- and 1/16-resolution levels keep the one-launch split-K path).  X, Y, Z: full resolution of the layer."""
    if not (dtype == torch.bfloat16 or (dtype == torch.float32 and x3 is True)):
        return False
    kc = conv_kc(dtype)
    if C0 < kc or C0 % kc or C1 < kc or C1 % kc or cout < 64 or cout % 64 or (X | Y | Z) & 1:
        return False
    bn = 256 if cout % 256 == 0 else 128 if cout % 128 == 0 else 64
    tx = 4 if bn == 256 else 8
    tiles = lambda x, y, z: B * (-(-x // tx)) * (-(-y // 8)) * (-(-z // 8)) * (cout // bn)
    return tiles(X // 2, Y // 2, Z // 2) * 8 >= 256 and tiles(X, Y, Z) >= 256


class PackBook:
    """The weight images of one model as persistent buffers.  ``get`` returns the image of a job (packing it on first use
    or when stale); after an optimizer step ``refresh`` rewrites EVERY image recorded so far in one launch
    (mmr_conv3d_k3_pack_batch) instead of one launch -- and, for the channel slices of the concat layers, one contiguous
    copy -- per image: 25 launches of a C3 training step become one.  Jobs read the layer's whole Keras kernel in place:
    rows row_off .. row_off + rows of its input-channel axis."""

    def __init__(self):
        self._e = {}

    def get(self, key, kind, w_keras, row_off, rows, mode):
        e = self._e.get(key)
        if e is None or e["w"].data_ptr() != w_keras.data_ptr() or e["job"] != (kind, int(row_off), int(rows), mode):
            _chk(w_keras, torch.float32, "w_keras")
            rows_total, cols = int(w_keras.shape[3]), int(w_keras.shape[4])
            nbytes = _lib.load().mmr_conv3d_k3_pack_job_bytes(kind, int(rows), cols, mode)
            if nbytes < 0:
                raise _lib.MmrError(f"cannot pack weights (kind {kind}, rows {rows} of {rows_total}, cols {cols}, mode {mode})")
            e = self._e[key] = dict(w=w_keras, job=(kind, int(row_off), int(rows), mode), dims=(rows_total, cols), valid=False,
                                    out=torch.empty(nbytes, dtype=torch.uint8, device=w_keras.device))
        if not e["valid"]:
            self._run([e])
        return e["out"]

    def invalidate(self):
        for e in self._e.values():
            e["valid"] = False

    def refresh(self):
        """Rewrite every stale image: one launch per arithmetic mode in the book (one, unless the backward is opt-in bf16)."""
        stale = [e for e in self._e.values() if not e["valid"]]
        for mode in sorted({e["job"][3] for e in stale}):
            self._run([e for e in stale if e["job"][3] == mode])

    @staticmethod
    def _run(entries):
        jobs = (_lib.PackJob * len(entries))()
        for j, e in zip(jobs, entries):
            kind, off, rows, _ = e["job"]
            j.w, j.out, j.kind, j.rows_total, j.row_off, j.rows, j.cols = (e["w"].data_ptr(), e["out"].data_ptr(), kind,
                                                                              e["dims"][0], off, rows, e["dims"][1])
        rc = _lib.load().mmr_conv3d_k3_pack_batch(jobs, len(entries), entries[0]["job"][3], _stream())
        _lib.check(rc, "mmr_conv3d_k3_pack_batch")
        for e in entries:
            e["valid"] = True


def pack_upfold_weights(w_keras, C0, dtype, x3=False):
    """Keras kernel [3,3,3,C0+C1,Cout] of a decoder layer -> (folded image of the C0 upsampled channels, plain image of the
    C1 skip channels)."""
    _chk(w_keras, torch.float32, "w_keras")
    cin, cout = int(w_keras.shape[3]), int(w_keras.shape[4])
    lib = _lib.load()
    mode = conv_mode(dtype, x3)
    nbytes = lib.mmr_conv3d_k3_upfold_packed_bytes(int(C0), cout, mode)
    if nbytes < 0:
        raise _lib.MmrError(f"cannot fold conv weights C0={C0} Cout={cout} for {dtype}")
    w_up = w_keras[:, :, :, :C0, :].contiguous()
    up = torch.empty(nbytes, dtype=torch.uint8, device=w_keras.device)
    rc = lib.mmr_conv3d_k3_upfold_pack(w_up.data_ptr(), up.data_ptr(), int(C0), cout, mode, _stream())
    _lib.check(rc, "mmr_conv3d_k3_upfold_pack")
    skip = pack_conv_weights(w_keras[:, :, :, C0:, :].contiguous(), dtype, x3=x3)
    return up, skip


def conv3d_k3_upfold(in_low, skip, w_up, w_skip, bias, cout, leaky=True, alpha=0.2, out_f32=False, x3=False, half_partial=None):
    """Conv3D(cout,3,'same')(concat([UpSampling3D(2)(in_low), skip])) + bias (+LeakyReLU) as two launches: the upsampled half
    on the low-resolution grid with folded weights (8 parity classes x 8 taps) into an fp32 partial tensor, then the skip
    half with its accumulators started from that partial."""
    dtype = in_low.dtype
    _chk(in_low, dtype, "in_low")
    _chk(skip, dtype, "skip")
    B, X2, Y2, Z2, C0 = in_low.shape
    X, Y, Z = 2 * X2, 2 * Y2, 2 * Z2
    if tuple(skip.shape[:4]) != (B, X, Y, Z):
        raise _lib.MmrError(f"skip shape {tuple(skip.shape)} does not match {(B, X, Y, Z)}")
    C1 = skip.shape[4]
    mode = conv_mode(dtype, x3)
    odt = torch.float32 if (out_f32 or dtype == torch.float32) else torch.bfloat16
    # the tensor between the two launches: IEEE half for bf16 layers (include/mmr.h), fp32 for fp32x3 layers
    half = (dtype == torch.bfloat16) if half_partial is None else bool(half_partial)
    partial = torch.empty((B, X, Y, Z, cout), dtype=torch.float16 if half else torch.float32, device=in_low.device)
    out = torch.empty((B, X, Y, Z, cout), dtype=odt, device=in_low.device)
    lib = _lib.load()
    fam = f"conv3d_k3_mfma_{('f32', 'bf16', 'f32x3', 'f32x1')[mode]}_bn{256 if cout % 256 == 0 else 128 if cout % 128 == 0 else 64}"
    # algorithmic flops = the plain 27-tap count of the layer's two halves; the folded half executes 8/27 of it
    with _Timed(fam + "_upfold", (C0, int(cout), X, Y, Z), 2.0 * 27 * C0 * cout * B * X * Y * Z):
        rc = lib.mmr_conv3d_k3_upfold_fwd(in_low.data_ptr(), C0, w_up.data_ptr(), partial.data_ptr(), int(half), B, X2, Y2, Z2,
                                          int(cout), mode, _stream())
    _lib.check(rc, "mmr_conv3d_k3_upfold_fwd")
    nws = lib.mmr_conv3d_k3_ksplit_ws_bytes(B, X, Y, Z, C1, int(cout), mode)   # > 0: the launch leaves a partial round of workgroups
    ws = _ws(nws, in_low.device) if nws > 0 else None
    with _Timed(fam + "_cinit", (C1, int(cout), X, Y, Z), 2.0 * 27 * C1 * cout * B * X * Y * Z):
        rc = lib.mmr_conv3d_k3_fwd_init(skip.data_ptr(), C1, w_skip.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        partial.data_ptr(), int(half), out.data_ptr(), B, X, Y, Z, int(cout), int(leaky),
                                        float(alpha), mode, int(out_f32), ws.data_ptr() if ws is not None else None, _stream())
    _lib.check(rc, "mmr_conv3d_k3_fwd_init")
    return out


def dgrad_upfold_supported(C0, C1, Cz, x3, B, X, Y, Z):
    """The folded data gradient (mmr_conv3d_k3_dgrad_upfold) serves the upsampled half of a concat layer's backward when the
    tensors are fp32 with bf16-split products (x3 True or 'hi'), the widths fit and both launches fill the chip.  X, Y, Z: full
    resolution of the layer."""
    if not x3 or C0 % 64 or C1 % 64 or C0 < 64 or C1 < 64 or Cz % 32 or (X | Y | Z) & 1:
        return False
    tiles = lambda x, y, z, c: B * (-(-x // (4 if c % 256 == 0 else 8))) * (-(-y // 8)) * (-(-z // 8)) * max(c // 256, 1)
    return tiles(X // 2, Y // 2, Z // 2, C0) >= 200 and tiles(X, Y, Z, C1) >= 200


def pack_dgrad_upfold_weights(w_keras, C0, x3=True):
    """Forward Keras kernel [3,3,3,C0+C1,Cz] of a concat layer -> operand image of the folded dgrad of its first C0 channels."""
    _chk(w_keras, torch.float32, "w_keras")
    cz = int(w_keras.shape[4])
    lib = _lib.load()
    mode = conv_mode(torch.float32, x3)
    nbytes = lib.mmr_conv3d_k3_dgrad_upfold_packed_bytes(cz, int(C0), mode)
    if nbytes < 0:
        raise _lib.MmrError(f"cannot fold dgrad weights C0={C0} Cz={cz}")
    w_up = w_keras[:, :, :, :C0, :].contiguous()
    out = torch.empty(nbytes, dtype=torch.uint8, device=w_keras.device)
    rc = lib.mmr_conv3d_k3_dgrad_upfold_pack(w_up.data_ptr(), out.data_ptr(), int(C0), cz, mode, _stream())
    _lib.check(rc, "mmr_conv3d_k3_dgrad_upfold_pack")
    return out


def conv3d_k3_dgrad_upfold(dz, w_packed, C0, ymask=None, dbias=None, alpha=0.2, accumulate=False, x3=True):
    """d(low-res input) of conv(concat([up2(x_low), skip])) given dz [B,X,Y,Z,Cz] -> [B,X/2,Y/2,Z/2,C0]; with ``ymask``
    (= x_low) the result is already multiplied by LeakyReLU'(x_low) and ``dbias`` (+)= its column sums."""
    _chk(dz, torch.float32, "dz")
    B, X, Y, Z, Cz = dz.shape
    X2, Y2, Z2 = X // 2, Y // 2, Z // 2
    out = torch.empty((B, X2, Y2, Z2, C0), dtype=torch.float32, device=dz.device)
    lib = _lib.load()
    mode = conv_mode(torch.float32, x3)
    ws = None
    if ymask is not None:
        _chk(ymask, torch.float32, "ymask")
        _chk(dbias, torch.float32, "dbias")
        if tuple(ymask.shape) != tuple(out.shape) or dbias.numel() != C0:
            raise _lib.MmrError(f"ymask {tuple(ymask.shape)} / dbias {tuple(dbias.shape)} do not match {tuple(out.shape)}")
        ws = _ws(lib.mmr_conv3d_k3_dgrad_upfold_ws_bytes(B, X2, Y2, Z2, int(C0)), dz.device)
    fam = f"conv3d_k3_mfma_{('f32', 'bf16', 'f32x3', 'f32x1')[mode]}_bn{256 if C0 % 256 == 0 else 128 if C0 % 128 == 0 else 64}_dgfold"
    # algorithmic flops: the 27-tap count of this half of the concat dgrad (the kernel executes 8/27 of it)
    with _Timed(fam, (Cz, int(C0), X, Y, Z), 2.0 * 27 * Cz * C0 * B * X * Y * Z):
        rc = lib.mmr_conv3d_k3_dgrad_upfold(dz.data_ptr(), Cz, w_packed.data_ptr(), out.data_ptr(), B, X2, Y2, Z2, int(C0),
                                            ymask.data_ptr() if ymask is not None else None, float(alpha),
                                            dbias.data_ptr() if dbias is not None else None,
                                            ws.data_ptr() if ws is not None else None, int(accumulate), mode, _stream())
    _lib.check(rc, "mmr_conv3d_k3_dgrad_upfold")
    return out


def dgrad_masked_pool_supported(cin, x3, X, Y, Z):
    """The masked data gradient can take the MaxPooling3D(2) backward of its output tensor into its epilogue
    (mmr_conv3d_k3_dgrad_masked_pool) on the 64-column fp32x3 / x1 tile with even volume dims."""
    return bool(x3) and cin % 64 == 0 and cin % 128 != 0 and not ((X | Y | Z) & 1)


def conv3d_k3_dgrad_masked(dz, wt_packed, cin, ymask, dbias, alpha=0.2, accumulate=False, x3=False, pool_grad=None):
    """d(input) of a k3 conv, already multiplied by LeakyReLU'(ymask) of the layer that produced that input, whose
    bias gradient (column sums of the result) lands in ``dbias``: the dgrad + leaky_bwd_bias_ pair in one kernel.
    ``pool_grad`` [B,X/2,Y/2,Z/2,cin]: the gradient that reaches the same tensor through MaxPooling3D(2) -- routed to each
    window's first maximum and added before the mask (maxpool3d2_bwd(masked=True) folded into the epilogue)."""
    _chk(dz, torch.float32, "dz")
    _chk(ymask, torch.float32, "ymask")
    _chk(dbias, torch.float32, "dbias")
    B, X, Y, Z, C0 = dz.shape
    if tuple(ymask.shape) != (B, X, Y, Z, cin) or dbias.numel() != cin:
        raise _lib.MmrError(f"ymask {tuple(ymask.shape)} / dbias {tuple(dbias.shape)} do not match {(B, X, Y, Z, cin)}")
    out = torch.empty((B, X, Y, Z, cin), dtype=torch.float32, device=dz.device)
    lib = _lib.load()
    ws = _ws(lib.mmr_conv3d_k3_dgrad_masked_ws_bytes(B, X, Y, Z, int(cin)), dz.device)
    mode = conv_mode(torch.float32, x3)
    fam = f"conv3d_k3_mfma_{('f32', 'bf16', 'f32x3', 'f32x1')[mode]}_bn{256 if cin % 256 == 0 else 128 if cin % 128 == 0 else 64 if cin % 64 == 0 else 32}"
    with _Timed(fam, (C0, int(cin), X, Y, Z), 2.0 * 27 * C0 * cin * B * X * Y * Z):
        if pool_grad is not None:
            _chk(pool_grad, torch.float32, "pool_grad")
            if tuple(pool_grad.shape) != (B, X // 2, Y // 2, Z // 2, cin):
                raise _lib.MmrError(f"pool_grad {tuple(pool_grad.shape)} != {(B, X // 2, Y // 2, Z // 2, cin)}")
            rc = lib.mmr_conv3d_k3_dgrad_masked_pool(dz.data_ptr(), C0, wt_packed.data_ptr(), out.data_ptr(), B, X, Y, Z, int(cin),
                                                     ymask.data_ptr(), float(alpha), dbias.data_ptr(), ws.data_ptr(),
                                                     int(accumulate), mode, pool_grad.data_ptr(), _stream())
        else:
            rc = lib.mmr_conv3d_k3_dgrad_masked(dz.data_ptr(), C0, wt_packed.data_ptr(), out.data_ptr(), B, X, Y, Z, int(cin),
                                                ymask.data_ptr(), float(alpha), dbias.data_ptr(), ws.data_ptr(),
                                                int(accumulate), mode, _stream())
    _lib.check(rc, "mmr_conv3d_k3_dgrad_masked")
    return out


def dgrad_split_supported(C0, C1, x3):
    """The split-store dgrad runs on the 16x16x32 kernels only (fp32x3 / x1, 64-column tiles)."""
    return bool(x3) and (C0 + C1) % 64 == 0 and C0 % 16 == 0 and C1 % 16 == 0


def conv3d_k3_dgrad_split(dz, wt_packed, C0, C1, y1=None, dbias1=None, alpha=0.2, accumulate=False, x3=True):
    """d(concat input) of a k3 conv stored split: (d0 [B,X,Y,Z,C0] compact, d1 [B,X,Y,Z,C1] times LeakyReLU'(y1) with
    dbias1 (+)= its column sums).  Callers gate on ``dgrad_split_supported`` (which mirrors the C-side check); a shape
    the kernel family does not cover raises instead of returning a half-updated state."""
    _chk(dz, torch.float32, "dz")
    B, X, Y, Z, Cz = dz.shape
    mode = conv_mode(torch.float32, x3)
    lib = _lib.load()
    d0 = torch.empty((B, X, Y, Z, C0), dtype=torch.float32, device=dz.device)
    d1 = torch.empty((B, X, Y, Z, C1), dtype=torch.float32, device=dz.device)
    ws = _ws(lib.mmr_conv3d_k3_dgrad_split_ws_bytes(B, X, Y, Z, C1), dz.device) if y1 is not None else None
    cin = C0 + C1
    fam = f"conv3d_k3_mfma_{('f32', 'bf16', 'f32x3', 'f32x1')[mode]}_bn{256 if cin % 256 == 0 else 128 if cin % 128 == 0 else 64 if cin % 64 == 0 else 32}"
    with _Timed(fam, (Cz, cin, X, Y, Z), 2.0 * 27 * Cz * cin * B * X * Y * Z):
        rc = lib.mmr_conv3d_k3_dgrad_split(dz.data_ptr(), Cz, wt_packed.data_ptr(), d0.data_ptr(), d1.data_ptr(), B, X, Y, Z,
                                           int(C0), int(C1), y1.data_ptr() if y1 is not None else None, float(alpha),
                                           dbias1.data_ptr() if y1 is not None else None,
                                           ws.data_ptr() if ws is not None else None, int(accumulate), mode, _stream())
    if rc == -3:  # MMR_EUNSUPPORTED: dgrad_split_supported() and the C gate disagree -- a bug, not a fallback case
        raise _lib.MmrError(f"mmr_conv3d_k3_dgrad_split does not cover C0={C0}, C1={C1}, Cz={Cz}, mode={mode}; "
                            "gate the call with ops.dgrad_split_supported()")
    _lib.check(rc, "mmr_conv3d_k3_dgrad_split")
    return d0, d1


def cin2_pool_supported(cout, out_dtype, x3=False):
    """The fused MaxPooling3D(2) epilogue exists in the two matrix-core first-layer kernels (bf16 out / fp32x3)."""
    mode = conv_mode(out_dtype, x3)
    # bf16: a 128-B output line holds 64 couts (the kernel writes whole lines); fp32x3: 32
    return (mode == BF16 and cout % 64 == 0 and cout <= 512) or (mode == F32X3 and cout % 32 == 0 and cout <= 320)


def conv3d_k3_cin2(src, trg, w_keras, bias, out_dtype, leaky=True, alpha=0.2, x3=False, pool=False):
    """First U-Net layer on concat([src, trg]) ([B,X,Y,Z,1] each, f32).  pool=True also returns MaxPooling3D(2) of the
    activated output from the same kernel (no separate pass over the full-resolution tensor) -> (out, pooled)."""
    _chk(src, torch.float32, "src")
    _chk(trg, torch.float32, "trg")
    _chk(w_keras, torch.float32, "w_keras")
    B, X, Y, Z = src.shape[:4]
    cout = int(w_keras.shape[-1])
    out = torch.empty((B, X, Y, Z, cout), dtype=out_dtype, device=src.device)
    pooled = torch.empty((B, X // 2, Y // 2, Z // 2, cout), dtype=out_dtype, device=src.device) if pool else None
    rc = _lib.load().mmr_conv3d_k3_cin2_fwd(src.data_ptr(), trg.data_ptr(), w_keras.data_ptr(),
                                            bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                            pooled.data_ptr() if pool else None,
                                            B, X, Y, Z, cout, int(leaky), float(alpha), conv_mode(out_dtype, x3), _stream())
    _lib.check(rc, "mmr_conv3d_k3_cin2_fwd")
    return (out, pooled) if pool else out


def flow_head_supported(cin, dtype, x3=False):
    """The folded-tap flow-head kernel keeps its weight image (Cin x 96 bf16, twice for fp32x3) plus at least one
    128-row P plane (41 KB) in LDS; wider inputs fall back to the generic MFMA conv."""
    if cin % 32 or (dtype == torch.float32 and not x3):
        return False
    return (2 if x3 else 1) * (cin // 8) * 96 * 16 + 128 * 81 * 4 <= 160 * 1024


def conv3d_k3_cout3(x, w_keras, bias, x3=False):
    """Flow head: Conv3D(3,3,'same') without activation, taps folded into the GEMM N axis -> fp32 [B,X,Y,Z,3]."""
    _chk(x, x.dtype, "x")
    _chk(w_keras, torch.float32, "w_keras")
    B, X, Y, Z, Cin = x.shape
    mode = conv_mode(x.dtype, x3)
    out = torch.empty((B, X, Y, Z, 3), dtype=torch.float32, device=x.device)
    with _Timed(f"flow_head_{('f32', 'bf16', 'f32x3')[mode]}", (Cin, 3, X, Y, Z), 2.0 * 27 * Cin * 3 * B * X * Y * Z):
        rc = _lib.load().mmr_conv3d_k3_cout3_fwd(x.data_ptr(), w_keras.data_ptr(),
                                                 bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                                 B, X, Y, Z, Cin, mode, _stream())
    _lib.check(rc, "mmr_conv3d_k3_cout3_fwd")
    return out


def maxpool3d2(x):
    _chk(x, x.dtype, "x")
    B, X, Y, Z, C = x.shape
    out = torch.empty((B, X // 2, Y // 2, Z // 2, C), dtype=x.dtype, device=x.device)
    rc = _lib.load().mmr_maxpool3d2_fwd(x.data_ptr(), out.data_ptr(), B, X, Y, Z, C, _DT[x.dtype], _stream())
    _lib.check(rc, "mmr_maxpool3d2_fwd")
    return out


# ------------------------------- losses --------------------------------- #
def _ws(nbytes, device):
    if nbytes < 0:
        raise _lib.MmrError("invalid workspace request")
    return torch.empty(max(int(nbytes), 8), dtype=torch.uint8, device=device)


def dice_loss(y_true, y_pred, return_parts=False, eps_mode=None, zeropad=False):
    """-mean_{b,l} ratio(2 sum(t*p), sum(t+p)); inputs [B,*S,L] f32 -> scalar tensor.
    eps_mode: 'divide_no_nan' | 'max_eps' (None = mmr.semantics default)."""
    _chk(y_true, torch.float32, "y_true")
    _chk(y_pred, torch.float32, "y_pred")
    if y_true.shape != y_pred.shape:
        raise _lib.MmrError("dice: shape mismatch")
    B, L = y_true.shape[0], y_true.shape[-1]
    nvox = y_true.numel() // (B * L)
    lib = _lib.load()
    ws = _ws(lib.mmr_dice_ws_bytes(B, nvox, L), y_true.device)
    loss = torch.empty(1, dtype=torch.float32, device=y_true.device)
    tb = torch.empty((B, L, 2), dtype=torch.float32, device=y_true.device)
    fn = lib.mmr_dice_zeropad_fwd_f32 if zeropad else lib.mmr_dice_fwd_f32
    if zeropad:   # losses.py:64-66 hard-codes tf.math.divide_no_nan whatever the voxelmorph version: the A6 switch is Dice's only
        eps_mode = "divide_no_nan"
    rc = fn(y_true.data_ptr(), y_pred.data_ptr(), loss.data_ptr(), tb.data_ptr(), ws.data_ptr(),
                              B, nvox, L, semantics.code("dice_eps", eps_mode), _stream())
    _lib.check(rc, "mmr_dice_fwd_f32")
    return (loss[0], tb) if return_parts else loss[0]


def dice_loss_bwd(y_true, top_bot, scale=1.0, out=None, eps_mode=None):
    """d dice_loss / d y_pred from the forward's (top, bot) sums; accumulates into ``out`` when given."""
    _chk(y_true, torch.float32, "y_true")
    _chk(top_bot, torch.float32, "top_bot")
    B, L = y_true.shape[0], y_true.shape[-1]
    nvox = y_true.numel() // (B * L)
    acc = out is not None
    if out is None:
        out = torch.empty_like(y_true)
    rc = _lib.load().mmr_dice_bwd_f32(y_true.data_ptr(), top_bot.data_ptr(), out.data_ptr(), B, nvox, L, float(scale),
                                      int(acc), semantics.code("dice_eps", eps_mode), _stream())
    _lib.check(rc, "mmr_dice_bwd_f32")
    return out


def grad_l2_loss(flow, loss_mult=1.0):
    _chk(flow, torch.float32, "flow")
    B, X, Y, Z, C = flow.shape
    lib = _lib.load()
    ws = _ws(lib.mmr_grad_l2_ws_bytes(B, X, Y, Z, C), flow.device)
    out = torch.empty(B, dtype=torch.float32, device=flow.device)
    rc = lib.mmr_grad_l2_fwd_f32(flow.data_ptr(), out.data_ptr(), ws.data_ptr(), B, X, Y, Z, C, float(loss_mult), _stream())
    _lib.check(rc, "mmr_grad_l2_fwd_f32")
    return out


_TICKETS = {}   # (device index, stream handle) -> zeroed int32 word: the in-kernel finalize of the loss reductions (csrc/losses.hip,
                # TicketFin) needs one per concurrently running call and leaves it zero; calls on one stream are ordered


def _ticket(device):
    key = (device.index, _stream())
    t = _TICKETS.get(key)
    if t is None:
        if len(_TICKETS) >= 64:
            _TICKETS.clear()
        t = _TICKETS[key] = torch.zeros(4, dtype=torch.int32, device=device)
    return t


def ncc_loss(I, J, win=9, eps=1e-5, form=None, out=None, scale=1.0):
    """vxm.losses.NCC(win, eps).loss -> [B]; form: 'classic' | 'clamped' (None = mmr.semantics default).
    ``out`` [B] given: out += scale * loss (a total loss assembled in place); the reduction is finished inside the kernel."""
    _chk(I, torch.float32, "I")
    _chk(J, torch.float32, "J")
    if I.shape != J.shape or I.shape[-1] != 1:
        raise _lib.MmrError("ncc: inputs must be [B,X,Y,Z,1] of equal shape")
    B, X, Y, Z, _ = I.shape
    lib = _lib.load()
    ws = _ws(lib.mmr_ncc_ws_bytes(B, X, Y, Z), I.device)
    acc = out is not None
    if acc:
        _chk(out, torch.float32, "out")
        if tuple(out.shape) != (B,):
            raise _lib.MmrError("ncc: out must be [B]")
    else:
        out = torch.empty(B, dtype=torch.float32, device=I.device)
    # the kernel mmr_ncc_fwd_f32 picks (csrc/losses.hip: ncc_fused4_ok): four z per lane for whole rows, else one
    four = Z % 4 == 0 and Z <= 256 and X * Y * Z * 4 < 0xF0000000
    with _Timed("hbm:ncc_fused4_kernel" if four else "hbm:ncc_fused_kernel", (X, Y, Z), 2.0 * 4 * B * X * Y * Z):   # I, J read once
        rc = lib.mmr_ncc_fwd_ticket_f32(I.data_ptr(), J.data_ptr(), out.data_ptr(), ws.data_ptr(), _ticket(I.device).data_ptr(),
                                        B, X, Y, Z, int(win), float(eps), semantics.code("ncc_form", form), float(scale), int(acc),
                                        _stream())
    _lib.check(rc, "mmr_ncc_fwd_ticket_f32")
    return out


def bending_energy(flow, out=None, scale=1.0):
    """Bending energy [B]; ``out`` [B] given: out += scale * energy (e.g. onto ``ncc_loss``'s result: two launches per step)."""
    _chk(flow, torch.float32, "flow")
    B, X, Y, Z, C = flow.shape
    if C != 3:
        raise _lib.MmrError("bending energy needs [B,X,Y,Z,3]")
    lib = _lib.load()
    ws = _ws(lib.mmr_bending_ws_bytes(B, X, Y, Z), flow.device)
    acc = out is not None
    if acc:
        _chk(out, torch.float32, "out")
        if tuple(out.shape) != (B,):
            raise _lib.MmrError("bending: out must be [B]")
    else:
        out = torch.empty(B, dtype=torch.float32, device=flow.device)
    with _Timed("hbm:bending_fused_kernel", (X, Y, Z), 3.0 * 4 * B * X * Y * Z):   # the field read once
        rc = lib.mmr_bending_fwd_ticket_f32(flow.data_ptr(), out.data_ptr(), ws.data_ptr(), _ticket(flow.device).data_ptr(),
                                            B, X, Y, Z, float(scale), int(acc), _stream())
    _lib.check(rc, "mmr_bending_fwd_ticket_f32")
    return out


def ncc_loss_bwd(I, J, gout=None, win=9, eps=1e-5, want=("I", "J"), form=None):
    """Gradients of ``ncc_loss`` [B] w.r.t. I and / or J, scaled by ``gout`` [B] (None = ones) -> (dI, dJ)."""
    _chk(I, torch.float32, "I")
    _chk(J, torch.float32, "J")
    if I.shape != J.shape or I.shape[-1] != 1:
        raise _lib.MmrError("ncc: inputs must be [B,X,Y,Z,1] of equal shape")
    B, X, Y, Z, _ = I.shape
    lib = _lib.load()
    ws = _ws(lib.mmr_ncc_bwd_ws_bytes(B, X, Y, Z), I.device)
    dI = torch.empty_like(I) if "I" in want else None
    dJ = torch.empty_like(J) if "J" in want else None
    if gout is not None:
        _chk(gout, torch.float32, "gout")
    rc = lib.mmr_ncc_bwd_f32(I.data_ptr(), J.data_ptr(), gout.data_ptr() if gout is not None else None,
                             dI.data_ptr() if dI is not None else None, dJ.data_ptr() if dJ is not None else None,
                             ws.data_ptr(), B, X, Y, Z, int(win), float(eps), semantics.code("ncc_form", form),
                             _stream())
    _lib.check(rc, "mmr_ncc_bwd_f32")
    return dI, dJ


def bending_energy_bwd(flow, gout=None, out=None):
    """Gradient of ``bending_energy`` [B] w.r.t. the field, scaled by ``gout``; accumulates into ``out`` when given."""
    _chk(flow, torch.float32, "flow")
    B, X, Y, Z, C = flow.shape
    if C != 3:
        raise _lib.MmrError("bending energy needs [B,X,Y,Z,3]")
    acc = out is not None
    if out is None:
        out = torch.empty_like(flow)
    rc = _lib.load().mmr_bending_bwd_f32(flow.data_ptr(), gout.data_ptr() if gout is not None else None, out.data_ptr(),
                                         B, X, Y, Z, int(acc), _stream())
    _lib.check(rc, "mmr_bending_bwd_f32")
    return out


# ------------------------------- generator ------------------------------ #
def philox_normal(shape, seed, stream_id=0, mean=0.0, std=1.0, device="cuda"):
    out = torch.empty(tuple(int(s) for s in shape), dtype=torch.float32, device=device)
    rc = _lib.load().mmr_philox_normal_f32(out.data_ptr(), out.numel(), int(seed) & (2 ** 64 - 1), int(stream_id),
                                           float(mean), float(std), _stream())
    _lib.check(rc, "mmr_philox_normal_f32")
    return out


def philox_uniform(shape, seed, stream_id=0, lo=0.0, hi=1.0, device="cuda"):
    out = torch.empty(tuple(int(s) for s in shape), dtype=torch.float32, device=device)
    rc = _lib.load().mmr_philox_uniform_f32(out.data_ptr(), out.numel(), int(seed) & (2 ** 64 - 1), int(stream_id),
                                            float(lo), float(hi), _stream())
    _lib.check(rc, "mmr_philox_uniform_f32")
    return out


def lut_u8(labels, lut256):
    _chk(labels, torch.uint8, "labels")
    _chk(lut256, torch.uint8, "lut")
    if lut256.numel() != 256:
        raise _lib.MmrError("lut must have 256 entries")
    out = torch.empty_like(labels)
    rc = _lib.load().mmr_lut_u8(labels.data_ptr(), out.data_ptr(), lut256.data_ptr(), labels.numel(), _stream())
    _lib.check(rc, "mmr_lut_u8")
    return out


def gmm_sample(labels, means, stds, seed=0, stream_id=0, noise=None):
    """labels u8 [B,...], means/stds f32 [B,L] -> image f32 of labels' shape."""
    _chk(labels, torch.uint8, "labels")
    _chk(means, torch.float32, "means")
    _chk(stds, torch.float32, "stds")
    B, L = means.shape
    nvox = labels.numel() // B
    if noise is not None:
        _chk(noise, torch.float32, "noise")
        if noise.numel() != labels.numel():
            raise _lib.MmrError("noise size mismatch")
    out = torch.empty(labels.shape, dtype=torch.float32, device=labels.device)
    rc = _lib.load().mmr_gmm_sample_f32(labels.data_ptr(), means.data_ptr(), stds.data_ptr(),
                                        noise.data_ptr() if noise is not None else None, out.data_ptr(), B, nvox, L,
                                        int(seed) & (2 ** 64 - 1), int(stream_id), _stream())
    _lib.check(rc, "mmr_gmm_sample_f32")
    return out


def blur_separable(x, kernels):
    """x f32 [B,X,Y,Z] (or [...,1]); kernels f32 [B,W] applied along x, y, z ('SAME' zero padding)."""
    _chk(x, torch.float32, "x")
    _chk(kernels, torch.float32, "kernels")
    B, X, Y, Z = x.shape[:4]
    W = kernels.shape[1]
    a, b = x, torch.empty_like(x)
    lib = _lib.load()
    for axis in range(3):
        rc = lib.mmr_blur_axis_f32(a.data_ptr(), b.data_ptr(), kernels.data_ptr(), B, X, Y, Z, axis, W, _stream())
        _lib.check(rc, "mmr_blur_axis_f32")
        a, b = b, (torch.empty_like(x) if axis == 0 else a)
    return a


def bias_clip_norm_gamma_(x, bias=None, gamma=None, lo=0.0, hi=255.0):
    """In place: x <- ((clip(x*exp(bias), lo, hi) - min_b)/(max_b - min_b)) ** exp(gamma[b])."""
    _chk(x, torch.float32, "x")
    B = x.shape[0]
    nvox = x.numel() // B
    lib = _lib.load()
    ws = _ws(lib.mmr_intensity_ws_bytes(B), x.device)
    rc = lib.mmr_bias_clip_norm_gamma_f32(x.data_ptr(), _chk(bias, torch.float32, "bias").data_ptr() if bias is not None else None,
                                          _chk(gamma, torch.float32, "gamma").data_ptr() if gamma is not None else None,
                                          ws.data_ptr(), B, nvox, float(lo), float(hi), _stream())
    _lib.check(rc, "mmr_bias_clip_norm_gamma_f32")
    return x


def onehot(labels, L):
    _chk(labels, torch.uint8, "labels")
    shp = labels.shape[:-1] if labels.shape[-1] == 1 else labels.shape
    out = torch.empty(tuple(shp) + (int(L),), dtype=torch.float32, device=labels.device)
    rc = _lib.load().mmr_onehot_f32(labels.data_ptr(), out.data_ptr(), labels.numel(), int(L), _stream())
    _lib.check(rc, "mmr_onehot_f32")
    return out


def argmax_u8(x):
    _chk(x, torch.float32, "x")
    C = x.shape[-1]
    out = torch.empty(x.shape[:-1], dtype=torch.uint8, device=x.device)
    rc = _lib.load().mmr_argmax_u8(x.data_ptr(), out.data_ptr(), out.numel(), C, _stream())
    _lib.check(rc, "mmr_argmax_u8")
    return out


def axpy_(y, x, a=1.0):
    _chk(y, torch.float32, "y")
    _chk(x, torch.float32, "x")
    if y.numel() != x.numel():
        raise _lib.MmrError("axpy size mismatch")
    rc = _lib.load().mmr_axpy_f32(y.data_ptr(), x.data_ptr(), float(a), y.numel(), _stream())
    _lib.check(rc, "mmr_axpy_f32")
    return y


# ------------------------------- training ------------------------------- #
def dice_labels_fwd(lab1, lab2, flow, L, zeropad=False, eps_mode=None):
    """Dice(one_hot(lab2), warp_linear(one_hot(lab1), flow)) from uint8 label maps [B,X,Y,Z(,1)]."""
    _chk(lab1, torch.uint8, "lab1")
    _chk(lab2, torch.uint8, "lab2")
    _chk(flow, torch.float32, "flow")
    B, X, Y, Z = flow.shape[:4]
    if lab1.numel() != B * X * Y * Z or lab2.numel() != lab1.numel():
        raise _lib.MmrError("label / flow shape mismatch")
    lib = _lib.load()
    ws = _ws(lib.mmr_dice_labels_ws_bytes(B, X * Y * Z, L), flow.device)
    loss = torch.empty(1, dtype=torch.float32, device=flow.device)
    tb = torch.empty((B, L, 2), dtype=torch.float32, device=flow.device)
    fn = lib.mmr_dice_labels_zeropad_fwd if zeropad else lib.mmr_dice_labels_fwd
    if zeropad:   # losses.py:64-66: always divide_no_nan
        eps_mode = "divide_no_nan"
    rc = fn(lab1.data_ptr(), lab2.data_ptr(), flow.data_ptr(), loss.data_ptr(), tb.data_ptr(),
            ws.data_ptr(), B, X, Y, Z, int(L), semantics.code("dice_eps", eps_mode), _stream())
    _lib.check(rc, "mmr_dice_labels_fwd")
    return loss[0], tb


def dice_labels_bwd(lab1, lab2, flow, top_bot, L, scale=1.0, out=None, zeropad=False, eps_mode=None):
    B, X, Y, Z = flow.shape[:4]
    acc = out is not None
    if out is None:
        out = torch.empty_like(flow)
    fn = _lib.load().mmr_dice_labels_zeropad_bwd if zeropad else _lib.load().mmr_dice_labels_bwd
    if zeropad:
        eps_mode = "divide_no_nan"
    rc = fn(lab1.data_ptr(), lab2.data_ptr(), flow.data_ptr(), top_bot.data_ptr(),
            out.data_ptr(), B, X, Y, Z, int(L), float(scale), int(acc), semantics.code("dice_eps", eps_mode), _stream())
    _lib.check(rc, "mmr_dice_labels_bwd")
    return out


def grad_l2_bwd(flow, loss_mult=1.0, scale=1.0, out=None):
    _chk(flow, torch.float32, "flow")
    B, X, Y, Z, C = flow.shape
    acc = out is not None
    if out is None:
        out = torch.empty_like(flow)
    rc = _lib.load().mmr_grad_l2_bwd_f32(flow.data_ptr(), out.data_ptr(), B, X, Y, Z, C, float(loss_mult), float(scale),
                                         int(acc), _stream())
    _lib.check(rc, "mmr_grad_l2_bwd_f32")
    return out


def resize_trilinear_bwd(dout, in_shape, mul=1.0, grid=None, zoom=0.0, separable=True):
    """Adjoint of resize_trilinear (same grid / zoom): dout [B,Xo,Yo,Zo,C] -> din [B,*in_shape,C].  ``separable`` (default): three
    per-axis passes through a work space; False: the one-launch 3-D gather (same result up to the summation order)."""
    _chk(dout, torch.float32, "dout")
    B, Xo, Yo, Zo, C = dout.shape
    X, Y, Z = (int(s) for s in in_shape)
    din = torch.empty((B, X, Y, Z, C), dtype=torch.float32, device=dout.device)
    lib = _lib.load()
    if separable:
        ws = _ws(lib.mmr_resize_trilinear_bwd_ws_bytes(B, X, Y, Z, C, Xo, Yo, Zo), dout.device)
        rc = lib.mmr_resize_trilinear_bwd_ws_f32(dout.data_ptr(), din.data_ptr(), ws.data_ptr(), B, X, Y, Z, C, Xo, Yo, Zo, float(mul),
                                                 semantics.code("resize_grid", grid), float(zoom), _stream())
        _lib.check(rc, "mmr_resize_trilinear_bwd_ws_f32")
        return din
    rc = lib.mmr_resize_trilinear_bwd_f32(dout.data_ptr(), din.data_ptr(), B, X, Y, Z, C, Xo, Yo, Zo, float(mul),
                                          semantics.code("resize_grid", grid), float(zoom), _stream())
    _lib.check(rc, "mmr_resize_trilinear_bwd_f32")
    return din


def compose_bwd(a, b, dout):
    _chk(dout, torch.float32, "dout")
    B, X, Y, Z, _ = a.shape
    da, db = torch.empty_like(a), torch.empty_like(b)
    rc = _lib.load().mmr_compose_bwd_f32(a.data_ptr(), b.data_ptr(), dout.data_ptr(), da.data_ptr(), db.data_ptr(), B, X, Y, Z, _stream())
    _lib.check(rc, "mmr_compose_bwd_f32")
    return da, db


def vecint_save(vel, nsteps):
    """-> (out, steps) with steps [max(nsteps-1,0), B,X,Y,Z,3] = inputs of squaring steps 1..n-1."""
    _chk(vel, torch.float32, "vel")
    B, X, Y, Z, _ = vel.shape
    out = torch.empty_like(vel)
    steps = torch.empty((max(nsteps - 1, 0),) + tuple(vel.shape), dtype=torch.float32, device=vel.device)
    rc = _lib.load().mmr_vecint_save_f32(vel.data_ptr(), steps.data_ptr() if nsteps > 1 else None, out.data_ptr(),
                                         B, X, Y, Z, int(nsteps), _stream())
    _lib.check(rc, "mmr_vecint_save_f32")
    return out, steps


def vecint_bwd(vel, steps, dout, nsteps):
    B, X, Y, Z, _ = vel.shape
    dvel, tmp = torch.empty_like(vel), torch.empty_like(vel)
    rc = _lib.load().mmr_vecint_bwd_f32(vel.data_ptr(), steps.data_ptr() if nsteps > 1 else None, dout.data_ptr(),
                                        dvel.data_ptr(), tmp.data_ptr(), B, X, Y, Z, int(nsteps), _stream())
    _lib.check(rc, "mmr_vecint_bwd_f32")
    return dvel


def warp3d_bwd(vol, flow, dout, need_vol=True, need_flow=True):
    """Gradients of warp3d(vol, flow, 'linear') -> (dvol | None, dflow | None)."""
    _chk(vol, torch.float32, "vol")
    _chk(flow, torch.float32, "flow")
    _chk(dout, torch.float32, "dout")
    B, X, Y, Z, C = vol.shape
    lib = _lib.load()
    dvol = dflow = None
    if need_flow:
        dflow = torch.empty_like(flow)
        _lib.check(lib.mmr_warp3d_bwd_flow_f32(vol.data_ptr(), flow.data_ptr(), dout.data_ptr(), dflow.data_ptr(),
                                               B, X, Y, Z, C, _stream()), "mmr_warp3d_bwd_flow_f32")
    if need_vol:
        dvol = torch.empty_like(vol)
        _lib.check(lib.mmr_warp3d_bwd_vol_f32(flow.data_ptr(), dout.data_ptr(), dvol.data_ptr(), B, X, Y, Z, C, _stream()),
                   "mmr_warp3d_bwd_vol_f32")
    return dvol, dflow


def leaky_bwd_bias_(y, dy, dbias, leaky=True, alpha=0.2, accumulate=False):
    """In place: dy <- dy * LeakyReLU'(y); dbias (+)= sum over voxels."""
    _chk(dy, torch.float32, "dy")
    C = dy.shape[-1]
    nvox = dy.numel() // C
    lib = _lib.load()
    ws = _ws(lib.mmr_leaky_bwd_ws_bytes(nvox, C), dy.device)
    rc = lib.mmr_leaky_bwd_bias_f32(y.data_ptr() if y is not None else None, dy.data_ptr(), dy.data_ptr(),
                                    dbias.data_ptr(), ws.data_ptr(), nvox, C, int(leaky), float(alpha), int(accumulate), _stream())
    _lib.check(rc, "mmr_leaky_bwd_bias_f32")
    return dy


def upcat_bwd(dcat, C0, C1, up0, d_in1=None, y0=None, dbias0=None, acc_b0=False, y1=None, dbias1=None, acc_b1=False,
              alpha=0.2):
    """Split d concat([up2(in0)|in0, in1]) -> (d_in0, d_in1); accumulates into d_in1 when given.  With y0 / y1 (the
    activated outputs of the layers that made in0 / in1) the parts come out multiplied by LeakyReLU'(y) and
    dbias0 / dbias1 (+)= their column sums (fused leaky_bwd_bias_)."""
    _chk(dcat, torch.float32, "dcat")
    B, X, Y, Z, C = dcat.shape
    s0 = (B, X // 2, Y // 2, Z // 2, C0) if up0 else (B, X, Y, Z, C0)
    d0 = torch.empty(s0, dtype=torch.float32, device=dcat.device)
    acc = d_in1 is not None
    if C1 > 0 and d_in1 is None:
        d_in1 = torch.empty((B, X, Y, Z, C1), dtype=torch.float32, device=dcat.device)
    lib = _lib.load()
    if y0 is None and y1 is None:
        rc = lib.mmr_upcat_bwd_f32(dcat.data_ptr(), d0.data_ptr(), d_in1.data_ptr() if C1 > 0 else None,
                                   B, X, Y, Z, C0, C1, int(up0), int(acc), _stream())
        _lib.check(rc, "mmr_upcat_bwd_f32")
        return d0, d_in1
    for t, shp, nm in ((y0, s0, "y0"), (y1, (B, X, Y, Z, C1), "y1")):
        if t is not None:
            _chk(t, torch.float32, nm)
            if tuple(t.shape) != tuple(shp):
                raise _lib.MmrError(f"{nm} shape {tuple(t.shape)} != {tuple(shp)}")
    ws = _ws(lib.mmr_upcat_bwd_masked_ws_bytes(C0, C1), dcat.device)
    rc = lib.mmr_upcat_bwd_masked_f32(dcat.data_ptr(), d0.data_ptr(), d_in1.data_ptr() if C1 > 0 else None, B, X, Y, Z, C0, C1,
                                      int(up0), int(acc), y0.data_ptr() if y0 is not None else None,
                                      y1.data_ptr() if y1 is not None else None, float(alpha),
                                      dbias0.data_ptr() if y0 is not None else None, int(acc_b0),
                                      dbias1.data_ptr() if y1 is not None else None, int(acc_b1), ws.data_ptr(), _stream())
    _lib.check(rc, "mmr_upcat_bwd_masked_f32")
    return d0, d_in1


def maxpool3d2_bwd(x, dpool, dx=None, masked=False, dbias=None, acc_b=False, alpha=0.2):
    """dx (+)= dpool routed to each window's first maximum; ``masked``: x is an activated LeakyReLU output, dx its
    pre-activation gradient -> routed values times LeakyReLU'(x), dbias (+)= their column sums."""
    _chk(x, torch.float32, "x")
    _chk(dpool, torch.float32, "dpool")
    B, X, Y, Z, C = x.shape
    acc = dx is not None
    if dx is None:
        dx = torch.empty_like(x)
    lib = _lib.load()
    if not masked:
        rc = lib.mmr_maxpool3d2_bwd_f32(x.data_ptr(), dpool.data_ptr(), dx.data_ptr(), B, X, Y, Z, C, int(acc), _stream())
        _lib.check(rc, "mmr_maxpool3d2_bwd_f32")
        return dx
    ws = _ws(lib.mmr_maxpool3d2_bwd_masked_ws_bytes(C), x.device)
    rc = lib.mmr_maxpool3d2_bwd_masked_f32(x.data_ptr(), dpool.data_ptr(), dx.data_ptr(), B, X, Y, Z, C, int(acc), 1,
                                           float(alpha), dbias.data_ptr(), int(acc_b), ws.data_ptr(), _stream())
    _lib.check(rc, "mmr_maxpool3d2_bwd_masked_f32")
    return dx


def conv3d_k3_wgrad(in0, dz, dw, in1=None, up0=False, accumulate=False, x3=False):
    """dw [3,3,3,C0+C1,Cout] (+)= weight gradient of conv3d_k3(in0, .., in1=in1, up0=up0) given dz."""
    _chk(in0, torch.float32, "in0")
    _chk(dz, torch.float32, "dz")
    _chk(dw, torch.float32, "dw")
    B, X, Y, Z, Cout = dz.shape
    C0 = in0.shape[-1]
    C1 = in1.shape[-1] if in1 is not None else 0
    lib = _lib.load()
    ws = _ws(lib.mmr_conv3d_k3_wgrad_ws_bytes(B, X, Y, Z, C0 + C1, Cout), dz.device)
    fn = lib.mmr_conv3d_k3_wgrad_f32x1 if x3 == "hi" else (lib.mmr_conv3d_k3_wgrad_f32x3 if x3 else lib.mmr_conv3d_k3_wgrad_f32)
    with _Timed("conv3d_k3_wgrad_mfma_" + ("f32x1" if x3 == "hi" else "f32x3" if x3 else "f32"), (C0 + C1, Cout, X, Y, Z),
                2.0 * 27 * (C0 + C1) * Cout * B * X * Y * Z):
        rc = fn(in0.data_ptr(), C0, int(up0), in1.data_ptr() if in1 is not None else None, C1,
                                         dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, X, Y, Z, Cout, int(accumulate), _stream())
    _lib.check(rc, "mmr_conv3d_k3_wgrad_f32")
    return dw


def wgrad_upfold_supported(C0, C1, Cout, x3, B, X, Y, Z):
    """The folded weight gradient (mmr_conv3d_k3_wgrad_upfold) serves a decoder layer conv(concat([up2(x_low), skip])) when the
    products are bf16 splits (x3 True or 'hi'), the widths fit, the tensors stay under the 3.75-GB buffer-descriptor range and
    the low-resolution volume has enough voxel tiles to keep 256 persistent workgroups busy.  X, Y, Z: full resolution."""
    if not x3 or C0 % 32 or C1 % 32 or C0 < 32 or C1 < 32 or Cout % 64 or (X | Y | Z) & 1:
        return False
    nv = B * X * Y * Z
    if nv * max(Cout, C1) * 4 > 0xF0000000 - 64:
        return False
    tiles_low = B * (-(-(X // 2) // 4)) * (-(-(Y // 2) // 8)) * (-(-(Z // 2) // 8))
    return tiles_low >= 256   # measured: at 250 low-res tiles (80^3 layer of C3) the folded pair is 0.1 ms slower than the one-launch kernel


def conv3d_k3_wgrad_upfold(x_low, skip, dz, dw, accumulate=False, x3=True):
    """dw [3,3,3,C0+C1,Cout] (+)= weight gradient of conv(concat([UpSampling3D(2)(x_low), skip])) given dz: the upsampled rows
    through the folded correlation on the low-resolution grid, the skip rows through the ordinary kernel."""
    _chk(x_low, torch.float32, "x_low")
    _chk(skip, torch.float32, "skip")
    _chk(dz, torch.float32, "dz")
    _chk(dw, torch.float32, "dw")
    B, X, Y, Z, Cout = dz.shape
    C0, C1 = x_low.shape[-1], skip.shape[-1]
    if tuple(x_low.shape[:4]) != (B, X // 2, Y // 2, Z // 2) or tuple(skip.shape[:4]) != (B, X, Y, Z):
        raise _lib.MmrError("wgrad_upfold: x_low / skip shapes do not match dz")
    if tuple(dw.shape) != (3, 3, 3, C0 + C1, Cout):
        raise _lib.MmrError(f"wgrad_upfold: dw {tuple(dw.shape)} != {(3, 3, 3, C0 + C1, Cout)}")
    lib = _lib.load()
    ws = _ws(lib.mmr_conv3d_k3_wgrad_upfold_ws_bytes(B, X // 2, Y // 2, Z // 2, C0, C1, Cout), dz.device)
    name = "conv3d_k3_wgrad_mfma_" + ("f32x1" if x3 == "hi" else "f32x3")
    # one timed family for the pair; algorithmic flops = the 27-tap count of the whole layer, executed = C1 rows + 8/27 of the C0 rows
    with _Timed(name + "_upfold", (C0 + C1, Cout, X, Y, Z), 2.0 * 27 * (C0 + C1) * Cout * B * X * Y * Z):
        rc = lib.mmr_conv3d_k3_wgrad_upfold(x_low.data_ptr(), C0, skip.data_ptr(), C1, dz.data_ptr(), dw.data_ptr(), ws.data_ptr(),
                                            B, X // 2, Y // 2, Z // 2, Cout, int(accumulate), 2 if x3 == "hi" else 1, _stream())
    _lib.check(rc, "mmr_conv3d_k3_wgrad_upfold")
    return dw


def conv3d_k3_cin2_wgrad(src, trg, dz, dw, accumulate=False, x3=False):
    B, X, Y, Z, Cout = dz.shape
    lib = _lib.load()
    ws = _ws(lib.mmr_conv3d_k3_cin2_wgrad_ws_bytes(Cout), dz.device)
    fn = lib.mmr_conv3d_k3_cin2_wgrad_f32x3 if x3 else lib.mmr_conv3d_k3_cin2_wgrad_f32
    rc = fn(src.data_ptr(), trg.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(),
                                          B, X, Y, Z, Cout, int(accumulate), _stream())
    _lib.check(rc, "mmr_conv3d_k3_cin2_wgrad_f32")
    return dw


def conv3d_k3_cout3_dgrad(dy, w_keras, x3=False):
    """Flow-head data gradient; ``x3``: bf16 hi/lo split products (Cin % 64 == 0), else exact fp32."""
    _chk(dy, torch.float32, "dy")
    B, X, Y, Z, _ = dy.shape
    Cin = w_keras.shape[3]
    dx = torch.empty((B, X, Y, Z, Cin), dtype=torch.float32, device=dy.device)
    lib = _lib.load()
    fn = lib.mmr_conv3d_k3_cout3_dgrad_f32x3 if (x3 and Cin % 64 == 0) else lib.mmr_conv3d_k3_cout3_dgrad_f32
    rc = fn(dy.data_ptr(), w_keras.data_ptr(), dx.data_ptr(), B, X, Y, Z, Cin, _stream())
    _lib.check(rc, "mmr_conv3d_k3_cout3_dgrad_f32")
    return dx


def conv3d_k3_cout3_dgrad_masked(dy, w_keras, ymask, dbias, alpha=0.2, accumulate=False, x3=False):
    """Flow-head dgrad with the producing layer's LeakyReLU backward + bias gradient fused (Cin % 64 == 0);
    returns None when the fused kernel does not cover the width (caller uses the unfused pair)."""
    _chk(dy, torch.float32, "dy")
    _chk(ymask, torch.float32, "ymask")
    B, X, Y, Z, _ = dy.shape
    Cin = w_keras.shape[3]
    if Cin % 64:
        return None
    if tuple(ymask.shape) != (B, X, Y, Z, Cin):
        raise _lib.MmrError(f"ymask {tuple(ymask.shape)} does not match {(B, X, Y, Z, Cin)}")
    dx = torch.empty((B, X, Y, Z, Cin), dtype=torch.float32, device=dy.device)
    lib = _lib.load()
    ws = _ws(lib.mmr_conv3d_k3_cout3_dgrad_masked_ws_bytes(B, X, Y, Z, Cin), dy.device)
    fn = lib.mmr_conv3d_k3_cout3_dgrad_masked_f32x3 if x3 else lib.mmr_conv3d_k3_cout3_dgrad_masked_f32
    rc = fn(dy.data_ptr(), w_keras.data_ptr(), dx.data_ptr(), B, X, Y, Z, Cin,
            ymask.data_ptr(), float(alpha), dbias.data_ptr(), ws.data_ptr(), int(accumulate), _stream())
    _lib.check(rc, "mmr_conv3d_k3_cout3_dgrad_masked_f32")
    return dx


def adam_step_(w, g, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
    for t, n in ((w, "w"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, torch.float32, n)
    rc = _lib.load().mmr_adam_step_f32(w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), w.numel(), float(lr),
                                       float(beta1), float(beta2), float(eps), int(step), float(grad_scale), _stream())
    _lib.check(rc, "mmr_adam_step_f32")
    return w
