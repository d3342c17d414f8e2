"""Host-side file helpers with the names the reference calls on ``vxm.py.utils`` / ``vxm.tf.utils``
(train_synthmorph.py:192,207; 3d_reg.py:323-326; bids_registration.py:339).

nibabel is not available here, so NIfTI-1 (.nii / .nii.gz) is read and written directly: 348-byte header,
little- or big-endian, scl_slope/scl_inter applied on load, sform (or qform / pixdim fallback) as the 4x4
affine.  FreeSurfer .mgh / .mgz (the format of the public SynthMorph label maps that ``load_labels`` is pointed at,
train_synthmorph.py:207) is read and written from its published layout as well.  .npy / .npz are supported as in
upstream ``load_volfile``.
"""
import gzip
import os
import struct

import numpy as np

_DT = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
       768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODE = {np.dtype(v).str[1:]: k for k, v in _DT.items()}


def _open(path, mode):
    return gzip.open(path, mode) if path.endswith(".gz") else open(path, mode)


def _quat_to_affine(hdr):
    b, c, d = hdr["quatern_b"], hdr["quatern_c"], hdr["quatern_d"]
    a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    pix = np.array(hdr["pixdim"][1:4], dtype=np.float64)
    qfac = -1.0 if hdr["pixdim"][0] < 0 else 1.0
    aff = np.eye(4)
    aff[:3, :3] = R * (pix * np.array([1, 1, qfac]))
    aff[:3, 3] = [hdr["qoffset_x"], hdr["qoffset_y"], hdr["qoffset_z"]]
    return aff


def read_nifti(path):
    """-> (data ndarray in file dtype scaled by slope/inter when set, affine 4x4, header dict)."""
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError(f"{path}: not a NIfTI-1 file")
    end = "<" if struct.unpack("<i", raw[:4])[0] == 348 else ">"
    if struct.unpack(end + "i", raw[:4])[0] != 348:
        raise ValueError(f"{path}: bad NIfTI-1 header size")
    dim = struct.unpack(end + "8h", raw[40:56])
    intent_code, datatype, bitpix = struct.unpack(end + "hhh", raw[68:74])
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "fff", raw[108:120])
    qform_code, sform_code = struct.unpack(end + "hh", raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(end + "6f", raw[256:280])
    srow = np.array(struct.unpack(end + "12f", raw[280:328]), dtype=np.float64).reshape(3, 4)
    if raw[344:347] not in (b"n+1", b"ni1"):
        raise ValueError(f"{path}: bad NIfTI-1 magic")
    hdr = dict(dim=dim, intent_code=intent_code, datatype=datatype, bitpix=bitpix, pixdim=pixdim, quatern_b=qb,
               quatern_c=qc, quatern_d=qd, qoffset_x=qx, qoffset_y=qy, qoffset_z=qz, qform_code=qform_code,
               sform_code=sform_code, scl_slope=slope, scl_inter=inter)
    if datatype not in _DT:
        raise ValueError(f"{path}: unsupported NIfTI datatype code {datatype}")
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    dt = np.dtype(_DT[datatype]).newbyteorder(end)
    off = int(vox_offset) if vox_offset >= 352 else 352
    data = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape)), offset=off).reshape(shape, order="F")
    data = data.astype(dt.newbyteorder("="))
    if slope not in (0.0, 1.0) or (inter != 0.0 and slope != 0.0):
        if np.isfinite(slope) and slope != 0.0:
            data = data.astype(np.float64) * slope + inter
    if sform_code > 0:
        aff = np.vstack([srow, [0, 0, 0, 1]])
    elif qform_code > 0:
        aff = _quat_to_affine(hdr)
    else:
        aff = np.diag(list(pixdim[1:4]) + [1.0]).astype(np.float64)
    return data, aff, hdr


def write_nifti(arr, path, affine=None, intent_code=0):
    arr = np.asarray(arr)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    key = arr.dtype.str[1:]
    if key not in _CODE:
        arr = arr.astype(np.float32)
        key = arr.dtype.str[1:]
    affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    nd = arr.ndim
    dim = [nd] + list(arr.shape) + [1] * (7 - nd)
    zooms = np.sqrt((affine[:3, :3] ** 2).sum(0))
    pixdim = [1.0] + list(zooms) + [1.0] * 4
    h = bytearray(348)
    struct.pack_into("<i", h, 0, 348)
    struct.pack_into("<8h", h, 40, *dim)
    struct.pack_into("<hhh", h, 68, int(intent_code), _CODE[key], arr.dtype.itemsize * 8)
    struct.pack_into("<8f", h, 76, *pixdim)
    struct.pack_into("<fff", h, 108, 352.0, 1.0, 0.0)
    h[123] = 2  # xyzt_units: mm
    struct.pack_into("<hh", h, 252, 0, 2)  # sform only (aligned)
    struct.pack_into("<12f", h, 280, *affine[:3].reshape(-1))
    h[344:348] = b"n+1\x00"
    with _open(path, "wb") as f:
        f.write(bytes(h))
        f.write(b"\x00\x00\x00\x00")
        f.write(np.asfortranarray(arr.astype(arr.dtype.newbyteorder("<"))).tobytes(order="F"))


# ---- FreeSurfer MGH / MGZ (version 1): big-endian; int32 version, width, height, depth, nframes, type, dof; int16 goodRASFlag;
# float32 spacing[3], x_ras[3], y_ras[3], z_ras[3] (direction cosines, column by column), c_ras[3] (world position of the
# volume centre); voxel data from byte 284, x fastest, frames last.  type: 0 uint8, 1 int32, 3 float32, 4 int16.
_MGH_DT = {0: ">u1", 1: ">i4", 3: ">f4", 4: ">i2"}
_MGH_CODE = {"u1": 0, "i4": 1, "f4": 3, "i2": 4}


def read_mgh(path):
    """-> (data ndarray [width, height, depth(, frames)], vox2ras affine 4x4, header dict)."""
    with (gzip.open(path, "rb") if path.endswith((".mgz", ".gz")) else open(path, "rb")) as f:
        raw = f.read()
    if len(raw) < 284:
        raise ValueError(f"{path}: not an MGH file")
    version, w, h, d, nf, typ, dof = struct.unpack(">7i", raw[:28])
    if version != 1 or typ not in _MGH_DT or min(w, h, d, nf) < 1:
        raise ValueError(f"{path}: unsupported MGH header (version {version}, type {typ}, dims {(w, h, d, nf)})")
    good = struct.unpack(">h", raw[28:30])[0]
    if good > 0:
        delta = np.array(struct.unpack(">3f", raw[30:42]), dtype=np.float64)
        mdc = np.array(struct.unpack(">9f", raw[42:78]), dtype=np.float64).reshape(3, 3).T   # columns x_ras, y_ras, z_ras
        c_ras = np.array(struct.unpack(">3f", raw[78:90]), dtype=np.float64)
    else:   # FreeSurfer's default orientation (coronal, LIA) when the flag is not set
        delta = np.ones(3)
        mdc = np.array([[-1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]])
        c_ras = np.zeros(3)
    dt = np.dtype(_MGH_DT[typ])
    n = w * h * d * nf
    if len(raw) < 284 + n * dt.itemsize:
        raise ValueError(f"{path}: truncated MGH data ({len(raw) - 284} of {n * dt.itemsize} bytes)")
    data = np.frombuffer(raw, dtype=dt, count=n, offset=284).reshape((w, h, d, nf), order="F").astype(dt.newbyteorder("="))
    if nf == 1:
        data = data[..., 0]
    M = mdc * delta
    aff = np.eye(4)
    aff[:3, :3] = M
    aff[:3, 3] = c_ras - M @ (np.array([w, h, d], dtype=np.float64) / 2.0)
    return data, aff, dict(dims=(w, h, d, nf), type=typ, dof=dof, goodRASFlag=good, delta=delta, Mdc=mdc, c_ras=c_ras)


def write_mgh(arr, path, affine=None):
    arr = np.asarray(arr)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    key = arr.dtype.str[1:]
    if key not in _MGH_CODE:
        arr = arr.astype(np.int32 if np.issubdtype(arr.dtype, np.integer) else np.float32)
        key = arr.dtype.str[1:]
    if arr.ndim == 3:
        arr = arr[..., None]
    if arr.ndim != 4:
        raise ValueError("MGH volumes are 3-D (+ frames)")
    w, h, d, nf = arr.shape
    affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    M = affine[:3, :3]
    delta = np.sqrt((M ** 2).sum(0))
    mdc = M / delta
    c_ras = affine[:3, 3] + M @ (np.array([w, h, d], dtype=np.float64) / 2.0)
    hd = bytearray(284)
    struct.pack_into(">7i", hd, 0, 1, w, h, d, nf, _MGH_CODE[key], 0)
    struct.pack_into(">h", hd, 28, 1)
    struct.pack_into(">3f", hd, 30, *delta)
    struct.pack_into(">9f", hd, 42, *mdc.T.reshape(-1))
    struct.pack_into(">3f", hd, 78, *c_ras)
    with (gzip.open(path, "wb") if path.endswith((".mgz", ".gz")) else open(path, "wb")) as f:
        f.write(bytes(hd))
        f.write(np.asfortranarray(arr.astype(arr.dtype.newbyteorder(">"))).tobytes(order="F"))


def load_volfile(filename, np_var="vol", add_batch_axis=False, add_feat_axis=False, pad_shape=None, resize_factor=1,
                 ret_affine=False):
    """``vxm.py.utils.load_volfile``: .nii / .nii.gz / .mgz / .mgh / .npy / .npz; squeezed; optional batch / feature axes."""
    if isinstance(filename, str) and not os.path.isfile(filename):
        raise ValueError("'%s' is not a file." % filename)
    affine = None
    if not isinstance(filename, str):
        vol = np.asarray(filename)
    elif filename.endswith((".nii", ".nii.gz")):
        vol, affine, _ = read_nifti(filename)
        vol = vol.squeeze()
    elif filename.endswith((".mgz", ".mgh", ".mgh.gz")):
        vol, affine, _ = read_mgh(filename)
        vol = vol.squeeze()
    elif filename.endswith(".npy"):
        vol = np.load(filename)
    elif filename.endswith(".npz"):
        npz = np.load(filename)
        vol = next(iter(npz.values())) if len(npz.keys()) == 1 else npz[np_var]
    else:
        raise ValueError("unknown filetype for %s" % filename)
    if pad_shape is not None or resize_factor != 1:
        raise NotImplementedError("pad_shape / resize_factor are not used by the reference")
    if add_feat_axis:
        vol = vol[..., np.newaxis]
    if add_batch_axis:
        vol = vol[np.newaxis, ...]
    return (vol, affine) if ret_affine else vol


def save_volfile(array, filename, affine=None):
    """``vxm.py.utils.save_volfile``: NIfTI (identity-like default affine) or .npz."""
    if filename.endswith((".nii", ".nii.gz")):
        if affine is None and array.ndim >= 3:  # upstream default: RAS-flipped diag + centring offset
            affine = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=float)
            pcrs = np.append(np.array(array.shape[:3]) / 2, 1)
            affine[:3, 3] = -np.matmul(affine, pcrs)[:3]
        write_nifti(array, filename, affine)
    elif filename.endswith((".mgz", ".mgh")):
        write_mgh(array, filename, affine)
    elif filename.endswith(".npz"):
        np.savez_compressed(filename, vol=array)
    elif filename.endswith(".npy"):
        np.save(filename, array)
    else:
        raise ValueError("unknown filetype for %s" % filename)


def load_labels(arg, ext=(".nii.gz", ".nii", ".mgz", ".npz", ".npy")):
    """``vxm.py.utils.load_labels``: every label map in a folder (or list / glob-free path) ->
    (sorted unique labels, list of uint arrays); shapes must agree."""
    if isinstance(arg, (tuple, list)):
        files = list(arg)
    elif os.path.isdir(arg):
        files = sorted(os.path.join(arg, f) for f in os.listdir(arg) if f.endswith(tuple(ext)))
    else:
        files = [arg]
    if not files:
        raise ValueError(f"no labels found for argument {arg!r}")
    maps = []
    shape = None
    for f in files:
        x = np.squeeze(load_volfile(f))
        if shape is None:
            shape = x.shape
        if not np.issubdtype(x.dtype, np.integer):
            raise ValueError(f"file {f!r} has non-integral data type")
        if x.shape != shape:
            raise ValueError(f"shape {x.shape} of file {f!r} is not {shape}")
        maps.append(x)
    return np.unique(maps), maps


def setup_device(gpuid=None):
    """``vxm.tf.utils.setup_device``: -> (device string, number of devices).  '-1' / None would mean CPU
    upstream; this engine has no CPU path and raises instead."""
    import torch
    if gpuid is None or str(gpuid) == "-1":
        raise RuntimeError("this engine runs on MI355X GPUs only (no CPU fallback)")
    ids = [g for g in str(gpuid).split(",") if g != ""]
    nb = len(ids)
    if torch.cuda.is_available() and nb > torch.cuda.device_count():
        raise RuntimeError(f"requested GPUs {ids} but only {torch.cuda.device_count()} visible")
    return "cuda:" + ids[0], nb
