"""``vxm.networks.VxmDense`` / ``Transform`` on MI355X.

Reference call sites: train_synthmorph.py:271-277,296-297; 3d_reg.py:277,
297-314,331-334; bids_two_steps_registration.py:303-325.  Semantics follow
SURVEY.md section 3.4 / Appendix A1.  The object keeps Keras' positional weight
list (``get_weights``/``set_weights``: [kernel (3,3,3,Cin,Cout), bias (Cout)] x
11 layers) so the reference's runtime-shape rebuild + weight transplant
(3d_reg.py:305-306) works unchanged.
"""
import json
import math
import types

import numpy as np
import torch

from . import ops
from .layers import d2h_volume, h2d_volume, to_device

DEFAULT_FEATURES = [[16, 32, 32, 32], [32, 32, 32, 32, 32, 16, 16]]


def _plan(enc, dec):
    """[(name, cin_parts, cout)] in Keras weight-creation order."""
    plan, skips = [], []
    cin = (2,)
    nlev = len(enc)
    for i, nf in enumerate(enc):
        plan.append((f"unet_enc_conv_{i}_0", cin, nf))
        skips.append(nf)
        cin = (nf,)
    for i in range(nlev):
        nf = dec[i]
        plan.append((f"unet_dec_conv_{i}_0", cin, nf))
        cin = (nf, skips.pop())
    for j, nf in enumerate(dec[nlev:]):
        plan.append((f"unet_dec_final_conv_{j}", cin, nf))
        cin = (nf,)
    plan.append(("flow", cin, 3))
    return plan


def _is_h5_name(path):
    return str(path).lower().endswith((".h5", ".hdf5", ".keras.h5"))


def _is_h5_file(path):
    with open(path, "rb") as fh:
        return fh.read(8) == b"\x89HDF\r\n\x1a\n"


def _attr_list(group, name):
    """Keras ``load_attributes_from_hdf5_group``: a string-array attribute, possibly split into name0, name1, ..."""
    a = group.attrs
    if name in a:
        vals = list(np.atleast_1d(a[name]))
    else:
        vals, i = [], 0
        while f"{name}{i}" in a:
            vals += list(np.atleast_1d(a[f"{name}{i}"]))
            i += 1
    return [v.decode("utf8") if isinstance(v, bytes) else str(v) for v in vals]


def _read_keras_weights(path):
    """-> (names of the layers that hold weights, [[arrays in weight_names order] per layer]) of a Keras .h5 file
    (full model: group 'model_weights'; save_weights: layer groups at the root)."""
    from . import h5lite
    with h5lite.File(path) as f:
        g = f
        if "layer_names" not in f.attrs and "layer_names0" not in f.attrs and "model_weights" in f:
            g = f["model_weights"]
        names, per_layer = [], []
        for ln in _attr_list(g, "layer_names"):
            lg = g[ln]
            wn = _attr_list(lg, "weight_names")
            if wn:
                names.append(ln)
                per_layer.append([np.asarray(lg[w][()]) for w in wn])
    return names, per_layer


def _write_keras_weights(g, layer_names, weights):
    g.attrs["layer_names"] = np.array([n.encode("utf8") for n in layer_names])
    g.attrs["backend"] = b"tensorflow"
    g.attrs["keras_version"] = b"2.7.0"
    for i, n in enumerate(layer_names):
        lg = g.create_group(n)
        wn = [f"{n}/kernel:0", f"{n}/bias:0"]
        lg.attrs["weight_names"] = np.array([w.encode("utf8") for w in wn])
        lg.create_dataset(wn[0], np.asarray(weights[2 * i], dtype=np.float32))
        lg.create_dataset(wn[1], np.asarray(weights[2 * i + 1], dtype=np.float32))


class InputModel:
    """The generator pair in front of the network: what ``tf.keras.Model(inputs=gen_model_1.inputs + gen_model_2.inputs,
    outputs=(ima_1, ima_2))`` is in train_synthmorph.py:290-295.  Calling it on two label-map batches runs both
    ``labels_to_image`` generators on the device and returns the two grayscale images; the label maps they were rendered
    from (after the generators' own nearest-neighbour warp) stay in ``.maps`` for the Dice term (``map_1`` / ``map_2``
    of :291-292), the one-hot tensors are only built when asked for."""

    def __init__(self, gen_1, gen_2):
        for g in (gen_1, gen_2):
            if not (hasattr(g, "generate") and hasattr(g, "in_shape") and hasattr(g, "L")):
                raise TypeError("InputModel needs two synth.labels_to_image generators")
        if gen_1.in_shape != gen_2.in_shape or gen_1.L != gen_2.L:
            raise ValueError("the two generators must share in_shape and the label list")
        self.gen_1, self.gen_2 = gen_1, gen_2
        self.inputs = list(gen_1.inputs) + list(gen_2.inputs)
        self.outputs = (gen_1.outputs[0], gen_2.outputs[0])
        self.maps = None

    def __call__(self, labels_1, labels_2, draws_1=None, draws_2=None, want_onehot=False):
        g1 = self.gen_1.generate(labels_1, draws=draws_1, want_onehot=want_onehot)
        g2 = self.gen_2.generate(labels_2, draws=draws_2, want_onehot=want_onehot)
        self.maps = (g1["onehot"] if want_onehot else g1["labels"], g2["onehot"] if want_onehot else g2["labels"])
        return g1["image"], g2["image"]


def _as_np(a):
    return a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


class VxmDense:
    """VoxelMorph dense registration network, forward on gfx950 HIP kernels.

    ``compute_dtype``: 'fp32x3' (default: fp32 tensors, every product as three bf16 MFMAs on hi/lo splits, measured
    3-5e-6 relative error -- inside the 1e-4 fp32 parity bar at 3/16 of the exact-fp32 time), 'fp32' (exact-fp32
    MFMA) or 'bf16' (bf16 activations/weights, fp32 accumulate: the throughput setting of BASELINE.json config 2).
    """

    def __init__(self, inshape, nb_unet_features=None, nb_unet_levels=None, unet_feat_mult=1,
                 nb_unet_conv_per_level=1, int_steps=7, svf_resolution=1, int_resolution=2,
                 int_downsize=None, bidir=False, use_probs=False, src_feats=1, trg_feats=1,
                 unet_half_res=False, input_model=None, hyp_model=None, fill_value=None,
                 reg_field="preintegrated", name="vxm_dense", compute_dtype="fp32x3", device="cuda", seed=0,
                 fold_upsampling=True):
        if len(inshape) != 3:
            raise ValueError("VxmDense here is 3-D only (the reference registers 3-D volumes)")
        if bidir or use_probs or unet_half_res or hyp_model is not None or nb_unet_conv_per_level != 1:
            raise NotImplementedError("bidir / use_probs / unet_half_res / hyp_model / conv_per_level!=1 "
                                      "are not used by the reference and not implemented")
        if src_feats != 1 or trg_feats != 1:
            raise NotImplementedError("src_feats = trg_feats = 1 only")
        if int_downsize is not None:
            int_resolution = int_downsize
        if nb_unet_features is None:
            nb_unet_features = DEFAULT_FEATURES
        enc, dec = [list(map(int, f)) for f in nb_unet_features]
        if len(dec) < len(enc):
            raise ValueError("decoder must have at least as many entries as the encoder")
        self.inshape = tuple(int(s) for s in inshape)
        self.enc, self.dec = enc, dec
        self.int_steps, self.int_resolution, self.svf_resolution = int(int_steps), int_resolution, svf_resolution
        self.fill_value = fill_value
        # decoder layers as folded upsampling (ops.conv3d_k3_upfold: the upsampled half of concat([up2(x), skip]) on the
        # low-resolution grid, 8/27 of its multiply-adds) wherever ops.upfold_supported says both launches fill the chip
        self.fold_upsampling = bool(fold_upsampling)
        self._packed_fold = {}
        self._book = ops.PackBook()   # every weight image of this model (forward here, the backward's in training.py)
        # input_model (train_synthmorph.py:294-296): its two outputs are source / target and its inputs (the label maps)
        # become the model's inputs.  Anything that is not the generator pair is refused rather than ignored.
        if input_model is not None and not isinstance(input_model, InputModel):
            raise TypeError("input_model must be a networks.InputModel(gen_1, gen_2) (the reference's generator pair, "
                            "train_synthmorph.py:288-296) or None")
        if input_model is not None and tuple(input_model.gen_1.in_shape) != tuple(int(s) for s in inshape):
            raise ValueError(f"input_model generates {tuple(input_model.gen_1.in_shape)} volumes, inshape is {tuple(inshape)}")
        self.input_model = input_model
        self.name = name
        self.device = torch.device(device)
        self.x3 = False
        if compute_dtype in ("bf16", torch.bfloat16):
            self.dtype = torch.bfloat16
        elif compute_dtype in ("fp32", "float32", torch.float32):
            self.dtype = torch.float32
        elif compute_dtype == "fp32x3":  # fp32 tensors, bf16 hi/lo split inside the convs (fp32-grade accuracy)
            self.dtype = torch.float32
            self.x3 = True
        else:
            raise ValueError("compute_dtype must be 'bf16', 'fp32' or 'fp32x3'")
        nlev = len(enc)
        if any(s % (2 ** nlev) for s in self.inshape):
            raise ValueError(f"inshape {self.inshape} must be divisible by 2**{nlev} (U-Net skip concat)")
        self._config = dict(inshape=list(self.inshape), nb_unet_features=[enc, dec], int_steps=self.int_steps,
                            svf_resolution=svf_resolution, int_resolution=int_resolution, fill_value=fill_value)
        self.plan = _plan(enc, dec)
        self._kc = ops.conv_kc(self.dtype)
        # Physical widths: every feature width is rounded up to the MFMA channel-slice size (64 bf16 / 32 fp32)
        # with zero weights and zero bias, so padded channels carry LeakyReLU(0) = 0 and results are unchanged
        # (e.g. voxelmorph's default [16,32,32,32] / [32,...,16,16] features run as 32/64-wide layers).
        ph = lambda c: -(-c // self._kc) * self._kc
        self.pplan = [(n, tuple(c if n == self.plan[0][0] else ph(c) for c in cin), cout if n == "flow" else ph(cout))
                      for n, cin, cout in self.plan]
        c0 = self.pplan[0][2]
        if self.dtype == torch.float32 and not self.x3 and not ((c0 <= 256 and 256 % c0 == 0) or c0 % 256 == 0):
            raise NotImplementedError("exact-fp32 first layer: padded width must divide 256 or be a multiple of 256")
        self._init_weights(seed)
        self.references = types.SimpleNamespace(
            unet_model=None, source=None, target=None, svf=None, preint_flow=None, postint_flow=None,
            pos_flow=None, neg_flow=None, y_source=None, y_target=None, hyp_input=None)
        self._losses = []

    # ------------------------------------------------------------------ weights
    def _logical_index(self, li):
        """(input-channel index into the physical kernel, logical cout) of layer li."""
        _, cin, cout = self.plan[li]
        _, pcin, _ = self.pplan[li]
        idx, off = [], 0
        for c, pc in zip(cin, pcin):
            idx += list(range(off, off + c))
            off += pc
        return torch.tensor(idx, device=self.device, dtype=torch.long), cout

    def _init_weights(self, seed):
        """All 22 arrays live in ONE flat fp32 buffer (views in Keras order, physical = padded widths) so that
        Adam and the data-parallel all-reduce are a single launch / a single collective."""
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        shapes = []
        for name_, cin, cout in self.pplan:
            shapes += [(3, 3, 3, sum(cin), cout), (cout,)]
        sizes = [int(np.prod(s)) for s in shapes]
        self._flat = torch.zeros(sum(sizes), dtype=torch.float32, device=self.device)
        self._w, off = [], 0
        for shp, n in zip(shapes, sizes):
            self._w.append(self._flat[off:off + n].view(shp))
            off += n
        init = []
        for i, (name_, cin, cout) in enumerate(self.plan):
            ci = sum(cin)
            if name_ == "flow":
                w = torch.randn((3, 3, 3, ci, cout), generator=g) * 1e-5  # RandomNormal(0, 1e-5)
            else:  # he_normal = truncated normal, stddev sqrt(2/fan_in)/0.8796
                std = math.sqrt(2.0 / (27 * ci)) / 0.87962566103423978
                w = torch.empty((3, 3, 3, ci, cout))
                torch.nn.init.trunc_normal_(w, 0.0, std, -2 * std, 2 * std, generator=g)
            init += [w, torch.zeros(cout)]
        self.set_weights(init)

    def get_weights(self):
        """22 arrays at the LOGICAL widths, Keras order (what ``set_weights`` of a rebuilt model accepts)."""
        out = []
        for li in range(len(self.plan)):
            idx, cout = self._logical_index(li)
            out.append(self._w[2 * li].index_select(3, idx)[..., :cout].detach().cpu().numpy())
            out.append(self._w[2 * li + 1][:cout].detach().cpu().numpy())
        return out

    def set_weights(self, weights):
        if len(weights) != len(self._w):
            raise ValueError(f"expected {len(self._w)} arrays, got {len(weights)}")
        staged = []
        for li, (name_, cin, cout) in enumerate(self.plan):
            k = to_device(weights[2 * li] if isinstance(weights[2 * li], torch.Tensor) else np.asarray(weights[2 * li]),
                          device=self.device)
            b = to_device(weights[2 * li + 1] if isinstance(weights[2 * li + 1], torch.Tensor)
                          else np.asarray(weights[2 * li + 1]), device=self.device)
            if tuple(k.shape) != (3, 3, 3, sum(cin), cout) or tuple(b.shape) != (cout,):
                raise ValueError(f"weight shapes {tuple(k.shape)}, {tuple(b.shape)} do not match layer {name_} "
                                 f"{(3, 3, 3, sum(cin), cout)}, {(cout,)}")
            staged.append((k, b))
        self._flat.zero_()
        for li, (k, b) in enumerate(staged):
            idx, cout = self._logical_index(li)
            self._w[2 * li][..., :cout].index_copy_(3, idx, k)
            self._w[2 * li + 1][:cout].copy_(b)
        self.invalidate_packed()

    def invalidate_packed(self):
        """Call after the flat parameter buffer was updated in place (optimizer step)."""
        self._packed = None
        self._packed_fold = {}
        self._book.invalidate()

    def repack(self):
        """After an optimizer step: every weight image made so far (forward, folded, the trainer's transposed ones)
        rewritten in place in one launch; the lists handed out by ``_pack`` stay valid."""
        self._book.invalidate()
        self._book.refresh()

    def _pack(self):
        if self._packed is None:
            mode = ops.conv_mode(self.dtype, self.x3)
            self._packed = [None] + [self._book.get(("fwd", i), ops.PACK_FWD, self._w[2 * i], 0, self._w[2 * i].shape[3], mode)
                                     for i in range(1, len(self.plan))]
        return self._packed

    def count_params(self):
        return sum(27 * sum(cin) * cout + cout for _, cin, cout in self.plan)

    def summary(self):
        print(f'Model: "{self.name}"  inshape={self.inshape}  compute_dtype={self.dtype}')
        for i, (n, cin, cout) in enumerate(self.plan):
            print(f"  {n:28s} Conv3D 3x3x3 {sum(cin):4d} -> {cout:4d}   params {27 * sum(cin) * cout + cout}")
        print(f"Total params: {self.count_params():,}")

    # ------------------------------------------------------------------ save / load
    def get_config(self):
        return dict(self._config)

    def _keras_names(self):
        """Keras layer names of the 11 weighted layers (vxm Unet 'unet_*' convs + '<name>_flow')."""
        return [f"{self.name}_flow" if n == "flow" else n for n, _, _ in self.plan]

    def save(self, path):
        """``.h5`` / ``.hdf5`` -> the Keras 2.x full-model HDF5 layout the reference's checkpoints use
        (train_synthmorph.py:313-317: root attrs keras_version / backend / model_config, group ``model_weights`` with
        ``layer_names`` and per-layer ``weight_names`` + datasets), readable by ``vxm.networks.VxmDense.load``;
        anything else -> safetensors."""
        if _is_h5_name(path):
            from . import h5lite
            root = h5lite.WGroup()
            cfg = {k: v for k, v in dict(self._config, name=self.name).items() if not (k == "fill_value" and v is None)}
            root.attrs["keras_version"] = b"2.7.0"
            root.attrs["backend"] = b"tensorflow"
            root.attrs["model_config"] = json.dumps({"class_name": "VxmDense", "config": cfg}).encode("utf8")
            _write_keras_weights(root.create_group("model_weights"), self._keras_names(), self.get_weights())
            h5lite.write_file(path, root)
            return
        from safetensors.torch import save_file
        tensors = {f"w{i:02d}": torch.from_numpy(np.ascontiguousarray(w)) for i, w in enumerate(self.get_weights())}
        save_file(tensors, path, metadata={"config": json.dumps(self._config), "format": "mmr-vxmdense-1"})

    def save_weights(self, path):
        """Keras ``save_weights``: the layer groups at the file root (no model_config)."""
        if not _is_h5_name(path):
            return self.save(path)
        from . import h5lite
        root = h5lite.WGroup()
        _write_keras_weights(root, self._keras_names(), self.get_weights())
        h5lite.write_file(path, root)

    def load_weights(self, path, by_name=False):
        """Positional load over the layers that have weights, like Keras' ``load_weights_from_hdf5_group``
        (``by_name=True`` matches layer names instead and leaves unmatched layers untouched)."""
        if _is_h5_file(path):
            names, per_layer = _read_keras_weights(path)
            mine = self._keras_names()
            if by_name:
                cur = self.get_weights()
                for n, ws in zip(names, per_layer):
                    if n in mine:
                        li = mine.index(n)
                        if len(ws) != 2:
                            raise ValueError(f"Layer {n} expects 2 weights, the file holds {len(ws)}")
                        cur[2 * li], cur[2 * li + 1] = ws
                self.set_weights(cur)
                return
            if len(per_layer) != len(mine):
                raise ValueError(f"You are trying to load a weight file containing {len(per_layer)} layers into a model "
                                 f"with {len(mine)} layers.")
            flat = []
            for n, ws in zip(names, per_layer):
                if len(ws) != 2:
                    raise ValueError(f"Layer {n} in the file holds {len(ws)} weight arrays, expected kernel + bias")
                flat += ws
            self.set_weights(flat)
            return
        from safetensors.torch import load_file
        t = load_file(path)
        self.set_weights([t[f"w{i:02d}"] for i in range(len(self._w))])

    @classmethod
    def load(cls, path, by_name=False, input_model=None, **kwargs):
        """``vxm.networks.VxmDense.load`` (3d_reg.py:277): constructor arguments from the file's model_config, then
        the weights.  Accepts Keras ``.h5`` files (detected by signature) and this package's safetensors files."""
        if _is_h5_file(path):
            from . import h5lite
            with h5lite.File(path) as f:
                raw = f.attrs.get("model_config")
            if raw is None:
                raise ValueError(f"{path}: no model_config attribute (a save_weights file? build the model and use load_weights)")
            cfg = json.loads(raw.decode("utf8") if isinstance(raw, bytes) else str(raw))
            if "config" not in cfg or "inshape" not in cfg["config"]:
                raise ValueError(f"{path}: model_config does not hold VxmDense constructor arguments "
                                 "(saved from a plain Keras functional model?)")
            cfg = dict(cfg["config"])
            cfg.pop("input_model", None)
        else:
            from safetensors import safe_open
            with safe_open(path, framework="pt") as f:
                cfg = json.loads(f.metadata()["config"])
        m = cls(input_model=input_model, **{**cfg, **kwargs})
        m.load_weights(path, by_name=by_name)
        return m

    # ------------------------------------------------------------------ forward
    def _conv(self, li, x, **kw):
        """Layer li of the plan on the MFMA kernel."""
        in1, cout = kw.get("in1"), self.pplan[li][2]
        if (self.fold_upsampling and kw.get("up0") and in1 is not None
                and ops.upfold_supported(x.shape[-1], in1.shape[-1], cout, self.dtype, self.x3, *in1.shape[:4])):
            if li not in self._packed_fold:
                wk, C0, mode = self._w[2 * li], int(x.shape[-1]), ops.conv_mode(self.dtype, self.x3)
                self._packed_fold[li] = (self._book.get(("upfold", li, C0), ops.PACK_UPFOLD, wk, 0, C0, mode),
                                         self._book.get(("skip", li, C0), ops.PACK_FWD, wk, C0, int(wk.shape[3]) - C0, mode))
            w_up, w_skip = self._packed_fold[li]
            return ops.conv3d_k3_upfold(x, in1, w_up, w_skip, self._w[2 * li + 1], cout, x3=self.x3,
                                        **{k: v for k, v in kw.items() if k not in ("in1", "up0")})
        return ops.conv3d_k3(x, self._packed[li], self._w[2 * li + 1], cout, x3=self.x3, **kw)

    def unet(self, src, trg):
        """[B,X,Y,Z,1] x2 (f32) -> flow [B,X,Y,Z,3] f32."""
        self._pack()
        w, nlev = self._w, len(self.enc)
        fuse = ops.cin2_pool_supported(int(w[0].shape[-1]), self.dtype, self.x3) and min(src.shape[1:4]) >= 2
        if fuse:   # MaxPooling3D(2) of the first layer comes out of its own epilogue
            last, pooled = ops.conv3d_k3_cin2(src, trg, w[0], w[1], self.dtype, x3=self.x3, pool=True)
        else:
            last, pooled = ops.conv3d_k3_cin2(src, trg, w[0], w[1], self.dtype, x3=self.x3), None
        skips = [last]
        li = 1
        for _ in range(1, nlev):
            last = pooled if pooled is not None else ops.maxpool3d2(last)
            pooled = None
            last = self._conv(li, last)
            skips.append(last)
            li += 1
        last = pooled if pooled is not None else ops.maxpool3d2(last)
        skip = None
        for _ in range(nlev):
            last = self._conv(li, last, in1=skip, up0=skip is not None)
            skip = skips.pop()
            li += 1
        for _ in self.dec[nlev:]:
            last = self._conv(li, last, in1=skip, up0=skip is not None)
            skip = None
            li += 1
        if skip is None and ops.flow_head_supported(last.shape[-1], self.dtype, self.x3):
            return ops.conv3d_k3_cout3(last, w[2 * li], w[2 * li + 1], x3=self.x3)  # taps folded into N
        return self._conv(li, last, in1=skip, up0=skip is not None, leaky=False, out_f32=True)

    def forward(self, source, target):
        """Device tensors [B,X,Y,Z,1] f32 -> dict(y_source, preint_flow, pos_flow, flow).  With an ``input_model`` the two
        arguments are the LABEL MAPS (uint8 [B,X,Y,Z,1]) and the generators' images become source / target."""
        if self.input_model is not None:
            source, target = self.input_model(source, target)
        if tuple(source.shape[1:4]) != self.inshape:
            raise ValueError(f"input shape {tuple(source.shape[1:4])} != model inshape {self.inshape}")
        flow = self.unet(source, target)
        svf = flow
        if self.svf_resolution != 1:
            svf = ops.rescale_transform(flow, 1.0 / self.svf_resolution)
        preint = svf
        if self.int_steps > 0 and self.int_resolution > 1 and self.int_resolution != self.svf_resolution:
            preint = ops.rescale_transform(svf, self.svf_resolution / self.int_resolution)
        pos = preint
        if self.int_steps > 0:
            pos = ops.vecint(preint, self.int_steps)
            if self.int_resolution > 1:
                pos = ops.resize_trilinear(pos, self.inshape, mul=float(self.int_resolution), pre_scale=True,
                                           zoom=float(self.int_resolution))
        y = ops.warp3d(source, pos, "linear", self.fill_value)
        r = self.references
        r.source, r.target, r.svf, r.preint_flow, r.postint_flow, r.pos_flow, r.y_source = \
            source, target, svf, preint, pos, pos, y
        return dict(y_source=y, preint_flow=preint, pos_flow=pos, flow=flow)

    __call__ = forward

    def predict(self, inputs, batch_size=None, verbose=0):
        """``model.predict([moving, fixed])`` -> [moved, preint_flow] as NumPy fp32 (3d_reg.py:310-314).  Keras predicts in
        batches; here one pair at a time bounds the activation memory, and for more than one pair the host <-> device copies
        of neighbouring pairs run on a second stream beside the forward (``_predict_overlapped``)."""
        src, trg = inputs
        n = np.asarray(src).shape[0] if not isinstance(src, torch.Tensor) else src.shape[0]
        host_in = not (isinstance(src, torch.Tensor) and src.is_cuda)
        if n > 1 and self.input_model is None and host_in and self.device.type == "cuda":
            return self._predict_overlapped(src, trg, n)
        out_m, out_f = [], []
        for b in range(n):
            if self.input_model is not None:   # label maps in: the generators take NumPy uint8 or device tensors as they are
                s, t = src[b:b + 1], trg[b:b + 1]
            elif host_in and self.device.type == "cuda":
                # both volumes pinned where they lie, converted to fp32 by a kernel that reads them over PCIe (hostio)
                from . import hostio
                s, t = hostio.pair_to_device([_as_np(src[b:b + 1]), _as_np(trg[b:b + 1])], self.device)
            else:
                s = h2d_volume(src[b:b + 1], self.device, tag=0)
                t = h2d_volume(trg[b:b + 1], self.device, tag=1)
            o = self.forward(s, t)
            if self.device.type == "cuda":
                from . import hostio
                ym, yf = hostio.many_to_host([o["y_source"], o["preint_flow"]])
            else:
                ym, yf = d2h_volume(o["y_source"], tag=0), d2h_volume(o["preint_flow"], tag=1)
            out_m.append(ym)
            out_f.append(yf)
        if n == 1:
            return [out_m[0], out_f[0]]
        return [np.concatenate(out_m), np.concatenate(out_f)]

    def _predict_overlapped(self, src, trg, n):
        """Pairs b - 1 / b / b + 1 in flight at once.  The four host arrays of the whole call (both inputs, both outputs) are pinned
        where they lie for its duration; on a side stream the cast kernel reads pair b + 1 over PCIe and the copy kernel writes
        pair b - 1's outputs straight into the result arrays while pair b is in the forward (events order every hand-over) -- by
        the copy engines (mmr_memcpy_async), not by kernels: nothing competes with the forward for compute units.  The host only
        enqueues.  Same kernels on the same inputs: results identical to the one-pair path; if the runtime refuses to
        pin an array, the pairs go through that path one by one."""
        import ctypes
        from . import _lib, hostio
        dev = self.device
        prep = []
        for a in (src, trg):
            a = _as_np(a)
            if a.dtype not in hostio._CODES:
                a = a.astype(np.float64)
            prep.append(np.ascontiguousarray(a))
        half = tuple(s // self.svf_resolution for s in self.inshape)
        out_m = hostio.exclusive_empty((n,) + self.inshape + (1,), np.float32)      # mappings of their own: safe to pin
        out_f = hostio.exclusive_empty((n,) + half + (3,), np.float32)
        # inputs are pinned in place only when they certainly own their pages (hostio.registrable: large volumes); otherwise each
        # pair is memcpy'd into pinned staging memory on the host while the previous pair is in the forward
        regs = [hostio.Registered(a).__enter__() for a in (prep[0], prep[1], out_m, out_f)]
        try:
            if not (regs[2].ok and regs[3].ok):
                res = [self.predict([prep[0][b:b + 1], prep[1][b:b + 1]]) for b in range(n)]
                return [np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])]
            main = torch.cuda.current_stream(dev)
            if getattr(self, "_copy_stream", None) is None:
                self._copy_stream = torch.cuda.Stream(device=dev)
            cs = self._copy_stream
            cs.wait_stream(main)
            lib = _lib.load()
            per = [a[0:1].size for a in prep]
            code = [hostio._CODES[a.dtype] for a in prep]

            tdt = {hostio.F64: torch.float64, hostio.F32: torch.float32, hostio.U8: torch.uint8, hostio.I16: torch.int16}

            def stage_in(b):
                # copy ENGINE into a device buffer of the caller's dtype, then a device-side cast (microseconds): a kernel that
                # reads PCIe for a millisecond would sit on every CU and keep the forward's 512-thread workgroups off them
                devs = []
                with torch.cuda.stream(cs):
                    for k in range(2):
                        raw = torch.empty((1,) + prep[k].shape[1:], dtype=tdt[code[k]], device=dev)
                        st = None
                        if regs[k].ok:
                            srcp = prep[k].ctypes.data + b * per[k] * prep[k].itemsize
                        else:
                            st = hostio.staging(per[k] * prep[k].itemsize, ("ovl", k, b & 1))
                            st.wait()                                    # the copy that last read this buffer (pair b - 2) is done
                            np.copyto(st.view(prep[k].dtype, (1,) + prep[k].shape[1:]), prep[k][b:b + 1])
                            srcp = st.ptr
                        _lib.check(lib.mmr_memcpy_async(raw.data_ptr(), ctypes.c_void_p(srcp), per[k] * prep[k].itemsize, 0, cs.cuda_stream),
                                   "mmr_memcpy_async")
                        if st is not None:
                            st.event = torch.cuda.Event()
                            st.event.record(cs)
                        if code[k] == hostio.F32:
                            d = raw
                        else:
                            d = torch.empty(raw.shape, dtype=torch.float32, device=dev)
                            _lib.check(lib.mmr_cast_to_f32(raw.data_ptr(), d.data_ptr(), per[k], code[k], cs.cuda_stream), "mmr_cast_to_f32")
                        d.record_stream(main)
                        devs.append(d)
                    ev = torch.cuda.Event()
                    ev.record(cs)
                return devs[0], devs[1], ev
            def copy_out(b, ym, yf, ev_c):
                with torch.cuda.stream(cs):
                    cs.wait_event(ev_c)
                    for tns, arr in ((ym, out_m), (yf, out_f)):
                        dst = arr.ctypes.data + b * arr[0:1].nbytes
                        _lib.check(lib.mmr_memcpy_async(ctypes.c_void_p(dst), tns.data_ptr(), tns.numel() * 4, 1, cs.cuda_stream),
                                   "mmr_memcpy_async")
                ym.record_stream(cs)
                yf.record_stream(cs)
            nxt = stage_in(0)
            prev = None
            for b in range(n):
                s, t, ev_in = nxt
                main.wait_event(ev_in)
                o = self.forward(s, t)                   # queued FIRST: whatever the copies below make the host wait for (a copy
                ev_c = torch.cuda.Event()                # to or from registered -- not hipHostMalloc'd -- memory may return only
                ev_c.record(main)                        # when it has run), the GPU already has this pair's forward to work on
                if b + 1 < n:
                    nxt = stage_in(b + 1)
                if prev is not None:
                    copy_out(*prev)                      # pair b - 1's outputs leave while pair b is in the forward
                prev = (b, o["y_source"].detach(), o["preint_flow"].detach(), ev_c)
            copy_out(*prev)
            cs.synchronize()
            main.synchronize()
        finally:
            torch.cuda.synchronize(dev)                   # nothing may still read or write the pages when they are unpinned
            for r in regs:
                r.__exit__(None, None, None)
        return [out_m, out_f]


class Transform:
    """``vxm.networks.Transform(inshape, interp_method=, rescale=None, nb_feats=1)`` (3d_reg.py:331-334).

    ``predict([vol, trf])``: trf is at ``inshape / rescale`` when ``rescale`` is given and is
    brought to full resolution with RescaleTransform first (also for rescale == 1)."""

    def __init__(self, inshape, affine=False, interp_method="linear", rescale=None, fill_value=None, nb_feats=1,
                 device="cuda"):
        if affine:
            raise NotImplementedError("affine Transform is not used by the reference")
        ops.interp_code(interp_method)
        self.inshape = tuple(int(s) for s in inshape)
        self.interp_method, self.rescale, self.fill_value, self.nb_feats = interp_method, rescale, fill_value, nb_feats
        self.device = torch.device(device)

    def predict(self, inputs, batch_size=None, verbose=0):
        vol, trf = inputs
        v = h2d_volume(vol, self.device, tag=2)
        t = h2d_volume(trf, self.device, tag=3)
        if v.dim() == 4:
            v = v[..., None].contiguous()
        if tuple(v.shape[1:4]) != self.inshape or v.shape[-1] != self.nb_feats:
            raise ValueError(f"volume shape {tuple(v.shape)} does not match Transform{self.inshape + (self.nb_feats,)}")
        if self.rescale is not None:
            t = ops.resize_trilinear(t, self.inshape, mul=float(self.rescale), pre_scale=self.rescale >= 1,
                                     zoom=float(self.rescale))
        return d2h_volume(ops.warp3d(v, t, self.interp_method, self.fill_value), tag=2)
