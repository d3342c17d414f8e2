"""SynthMorph training step on MI355X: generators -> VxmDense forward -> Dice + Grad-l2 ->
hand-written backward -> (RCCL all-reduce) -> Adam.

Behaviour reproduced (train_synthmorph.py:284-344):
  * per-replica loss vector [b] = (Dice(map_2, pred) + 1) + Grad('l2', reg_param)(flow)[b];
    Keras differentiates its SUM and scales by 1/num_replicas, the replicas' gradients are
    SUM-all-reduced; the logged loss is the vector's mean;
  * Adam(lr) with Keras defaults beta1 .9, beta2 .999, eps 1e-7;
  * ModelCheckpoint('{epoch:04d}') every save_freq epochs, initial save, init_weights / init_epoch.
The autograd graph is not used: the backward pass is an explicit reverse walk over a tape of
kernel launches (all HIP, fp32).  One process per GPU; batch sharded by rank.
"""
import os
import time

import numpy as np
import torch

from . import ops, parallel
from .layers import to_device


class Adam:
    """Keras-style Adam over ONE flat parameter buffer (single fused launch)."""

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta_1, beta_2, epsilon
        self.t = 0
        self.m = self.v = None

    def apply(self, flat_w, flat_g, grad_scale=1.0):
        if self.m is None:
            self.m, self.v = torch.zeros_like(flat_w), torch.zeros_like(flat_w)
        self.t += 1
        ops.adam_step_(flat_w, flat_g, self.m, self.v, self.t, self.lr, self.b1, self.b2, self.eps, grad_scale)

    def state_dict(self):
        return {"t": self.t, "m": self.m, "v": self.v}


class SynthMorphTrainer:
    """model: fp32 VxmDense; gen_1/gen_2: synth.LabelsToImage (sharing the label list)."""

    def __init__(self, model, gen_1=None, gen_2=None, reg_param=1.0, optimizer=None, zero_pad_dice=False,
                 process_group=None, world_size=1, rank=0, backward_precision=None, overlap_wgrad=False, fuse_pool_bwd=True,
                 batch_repack=True, early_reduce=True):
        """gen_1 / gen_2 default to the generator pair of ``model.input_model`` (a model built the reference's way,
        ``VxmDense(..., input_model=InputModel(gen_1, gen_2))``, train_synthmorph.py:294-296).
        backward_precision: None = same arithmetic as the forward (fp32 / fp32x3); 'bf16' = dgrad and wgrad
        products on the bf16 hi halves only (one MFMA instead of three, fp32 accumulate) -- an opt-in
        mixed-precision backward; the forward (and therefore every output and loss) keeps fp32-grade accuracy.
        overlap_wgrad (opt-in; True, or "small" = only the levels of at most 100 k voxels, whose launches underfill the chip):
        the weight-gradient kernels run on a second HIP stream (nothing on the data-gradient chain
        reads a weight gradient before Adam).  Measured at C3: 26.0 -> 25.9 ms -- the matrix-core kernels hold every CU's LDS
        and both waves' full register budget per SIMD, so the HBM-bound kernels of the chain cannot run beside them and only
        the partial last rounds of launches fill; off by default so that per-kernel timings stay those of a kernel alone on
        the chip.  Same kernels, same results.
        fuse_pool_bwd: the MaxPooling3D backward of a skip tensor runs in the epilogue of the decoder conv's data gradient
        (ops.conv3d_k3_dgrad_masked(pool_grad=)) where that kernel supports it; False keeps the separate pooling-backward pass.
        batch_repack: after the optimizer step every weight image of the model (forward, folded, transposed) is rewritten by
        one launch (VxmDense.repack); False marks them stale and each is packed by its own launch where the next step needs it.
        early_reduce: data parallel only -- True (default) all-reduces everything but the encoder's gradients asynchronously
        from the point where the backward reaches the last encoder conv (two buckets); False = ONE all-reduce of the whole flat
        buffer after the backward (bench.py --no-early-reduce: the A/B the first multi-GPU run needs)."""
        if model.dtype != torch.float32:
            raise NotImplementedError("training runs the fp32 path (the reference trains in fp32)")
        im = getattr(model, "input_model", None)
        if gen_1 is None and gen_2 is None and im is not None:
            gen_1, gen_2 = im.gen_1, im.gen_2
        if gen_1 is None or gen_2 is None:
            raise ValueError("SynthMorphTrainer needs the two labels_to_image generators (arguments or model.input_model)")
        if im is not None and (im.gen_1 is not gen_1 or im.gen_2 is not gen_2):
            raise ValueError("the generators passed differ from model.input_model's")
        self.model, self.gen_1, self.gen_2 = model, gen_1, gen_2
        self.L = gen_1.L
        self.reg_param = float(reg_param)
        self.opt = optimizer or Adam(1e-4)
        self.zero_pad_dice = zero_pad_dice
        if backward_precision not in (None, "bf16"):
            raise ValueError("backward_precision must be None or 'bf16'")
        self.bwd_x3 = "hi" if backward_precision == "bf16" else model.x3
        self.pg, self.world, self.rank = process_group, int(world_size), int(rank)
        self.fuse_pool_bwd = bool(fuse_pool_bwd)
        self.batch_repack = bool(batch_repack)
        self.wstream = torch.cuda.Stream(device=model._flat.device) if (overlap_wgrad and model._flat.is_cuda) else None
        # "small": only the levels whose launches leave most of the chip idle (<= 40^3 voxels of a 160^3 step)
        self.wstream_max_voxels = 100_000 if overlap_wgrad == "small" else None
        self.gstream, self._ahead = None, None   # generator stream and the pair it rendered ahead (train_step(next_labels=))
        self.render_at = "end"   # where a step queues the next pair's renderings: behind the whole step ("end") or behind the
        # flow head, beside the small kernels of the tail ("tail"); measured equal (24.3 ms both, 24.7 without)
        self.gflat = torch.zeros_like(model._flat)
        self.g, self.goff, off = [], [], 0
        for w in model._w:
            n = w.numel()
            self.g.append(self.gflat[off:off + n].view(w.shape))
            self.goff.append(off)
            off += n
        # data parallel: the gradients of every layer behind the encoder are complete when the backward reaches the last encoder
        # conv -- their all-reduce starts there and runs under the encoder's backward (bucket 1); the rest follows the backward
        self._bucket_li = len(model.enc) - 1
        self._ar_early, self._reduce_early = None, False   # only train_step reduces; forward_backward leaves gflat unreduced
        self.early_reduce = bool(early_reduce)
        # HIP events around the encoder's part of the backward (from the point where bucket 1 would start to the end of the
        # walk) when time_encoder_bwd is set: [(start, end)] per step -- does the in-flight collective slow those kernels?
        self.time_encoder_bwd, self.encoder_bwd_events = False, []
        if self.world > 1 or parallel.forced():
            parallel.broadcast_(model._flat, 0, self.pg)
            model.invalidate_packed()

    # ------------------------------------------------------------------ forward with tape
    def _forward(self, src, trg, tape):
        m = self.model
        m._pack()
        w, nlev = m._w, len(m.enc)
        fuse = ops.cin2_pool_supported(int(w[0].shape[-1]), torch.float32, m.x3) and min(src.shape[1:4]) >= 2
        if fuse:   # the first MaxPooling3D(2) comes out of the first layer's epilogue
            last, pooled = ops.conv3d_k3_cin2(src, trg, w[0], w[1], torch.float32, x3=m.x3, pool=True)
        else:
            last, pooled = ops.conv3d_k3_cin2(src, trg, w[0], w[1], torch.float32, x3=m.x3), None
        tape.append(("conv0", 0, src, trg, last))
        skips = [last]
        li = 1

        def conv(x, in1=None, up0=False, leaky=True, cout=None):
            nonlocal li
            if cout == 3 and in1 is None and not up0 and ops.flow_head_supported(x.shape[-1], torch.float32, m.x3):
                y = ops.conv3d_k3_cout3(x, w[2 * li], w[2 * li + 1], x3=True)
                tape.append(("conv", li, x, up0, in1, y, leaky))
                li += 1
                return y
            if up0 and in1 is not None and cout is None:
                # decoder layers: folded upsampling where it applies (VxmDense._conv decides); the tape records the same
                # (inputs, output) either way, so the backward does not care how the forward was computed
                y = m._conv(li, x, in1=in1, up0=True, leaky=leaky, out_f32=True)
            else:
                y = ops.conv3d_k3(x, m._packed[li], w[2 * li + 1], m.pplan[li][2] if cout is None else cout, in1=in1,
                                  up0=up0, leaky=leaky, out_f32=True, x3=m.x3)
            tape.append(("conv", li, x, up0, in1, y, leaky))
            li += 1
            return y

        for _ in range(1, nlev):
            p = pooled if pooled is not None else ops.maxpool3d2(last)
            pooled = None
            tape.append(("pool", last, p))
            last = conv(p)
            skips.append(last)
        p = pooled if pooled is not None else ops.maxpool3d2(last)
        tape.append(("pool", last, p))
        last, skip = p, None
        for _ in range(nlev):
            last = conv(last, in1=skip, up0=skip is not None)
            skip = skips.pop()
        for _ in m.dec[nlev:]:
            last = conv(last, in1=skip, up0=skip is not None)
            skip = None
        return conv(last, in1=skip, up0=skip is not None, leaky=False, cout=3)

    def _tail_forward(self, flow):
        m = self.model
        half = tuple(s // m.svf_resolution for s in m.inshape)
        svf = ops.resize_trilinear(flow, half, mul=1.0 / m.svf_resolution, zoom=1.0 / m.svf_resolution) \
            if m.svf_resolution != 1 else flow
        if m.int_resolution != m.svf_resolution:
            raise NotImplementedError("training path assumes int_resolution == svf_resolution (both shipped configs)")
        pos_lo, steps = ops.vecint_save(svf, m.int_steps)
        pos = ops.resize_trilinear(pos_lo, m.inshape, mul=float(m.int_resolution), pre_scale=True,
                                   zoom=float(m.int_resolution)) \
            if m.int_resolution != 1 else pos_lo
        return svf, steps, pos_lo, pos

    def _tail_backward(self, dpos, svf, steps):
        m = self.model
        half = tuple(svf.shape[1:4])
        dpos_lo = ops.resize_trilinear_bwd(dpos, half, mul=float(m.int_resolution), zoom=float(m.int_resolution)) \
            if m.int_resolution != 1 else dpos
        dsvf = ops.vecint_bwd(svf, steps, dpos_lo, m.int_steps)
        if m.svf_resolution != 1:
            return ops.resize_trilinear_bwd(dsvf, m.inshape, mul=1.0 / m.svf_resolution, zoom=1.0 / m.svf_resolution)
        return dsvf

    # ------------------------------------------------------------------ backward over the tape
    def _backward(self, tape, dflow):
        m = self.model
        bmode = ops.conv_mode(torch.float32, self.bwd_x3)
        grads = {}
        first = tape[-1]
        grads[id(first[5])] = dflow

        # activated conv outputs -> index of the layer that made them.  A gradient stored for such a tensor is kept
        # "pre-masked" where possible: every kernel that produces a contribution multiplies it by LeakyReLU'(y) and adds
        # its column sums to that layer's bias gradient (both are linear in the contribution), so no separate
        # leaky-backward pass runs over it.
        act = {id(r[5]): r[1] for r in tape if r[0] == "conv" and r[6]}
        act.update({id(r[4]): 0 for r in tape if r[0] == "conv0"})
        premasked, bias_started = set(), set()

        def want_mask(t):
            return (t is not None and id(t) in act and t.shape[-1] % 4 == 0
                    and (id(t) not in grads or id(t) in premasked))

        def bias_of(t):
            """(bias-gradient view, accumulate?) of the layer that produced t; the first contribution writes."""
            l = act[id(t)]
            acc = l in bias_started
            bias_started.add(l)
            return self.g[2 * l + 1], acc

        def add_grad(t, g):
            if id(t) in grads:
                ops.axpy_(grads[id(t)], g, 1.0)
            else:
                grads[id(t)] = g

        # A skip tensor that also feeds a MaxPooling3D(2) gets two gradient contributions: the data gradient of the decoder layer
        # that reads it and the pooling's.  Where ops.dgrad_masked_pool_supported says so, the first one is DEFERRED until the
        # second exists and the pooling backward then runs in that conv's epilogue (no pass that reads the tensor and reads +
        # rewrites its whole gradient: 3.3 GB at the first level of a 160^3 step).
        pooled_from = {id(r[1]) for r in tape if r[0] == "pool"}
        deferred = {}
        side = self.wstream

        def wgrad(fn, dz):
            """Run the weight-gradient launch ``fn`` behind everything enqueued so far, on the side stream when there is one.
            ``dz`` is a temporary of the main stream that the side stream reads: the allocator must not hand its block out again
            before that read has run (record_stream); activations live on the tape until the streams have been joined."""
            if side is None or (self.wstream_max_voxels is not None
                                and dz.shape[1] * dz.shape[2] * dz.shape[3] > self.wstream_max_voxels):
                return fn()
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                fn()
            dz.record_stream(side)

        enc_ev0 = None
        for rec in reversed(tape):
            kind = rec[0]
            if kind == "conv":
                _, li, x, up0, in1, y, leaky = rec
                if li == self._bucket_li and self.time_encoder_bwd:
                    enc_ev0 = torch.cuda.Event(enable_timing=True)
                    enc_ev0.record()
                if (self._reduce_early and li == self._bucket_li and self._bucket_li > 0 and side is None
                        and (self.world > 1 or parallel.forced())):
                    # every layer > li has its weight and bias gradient by now (a layer's bias sums are closed before its own
                    # record is reached): all-reduce them behind what is queued, under the rest of the backward
                    self._ar_early = parallel.allreduce_sum_async(self.gflat[self.goff[2 * (li + 1)]:], self.pg)
                dy = grads.pop(id(y))
                if id(y) in premasked:
                    dz = dy
                else:
                    dz = ops.leaky_bwd_bias_(y if leaky else None, dy, self.g[2 * li + 1], leaky=leaky)
                if (up0 and in1 is not None and m.fold_upsampling
                        and ops.wgrad_upfold_supported(x.shape[-1], in1.shape[-1], dz.shape[-1], self.bwd_x3, *dz.shape[:4])):
                    # upsampled rows on the low-res grid
                    wgrad(lambda: ops.conv3d_k3_wgrad_upfold(x, in1, dz, self.g[2 * li], x3=self.bwd_x3), dz)
                else:
                    wgrad(lambda: ops.conv3d_k3_wgrad(x, dz, self.g[2 * li], in1=in1, up0=up0, x3=self.bwd_x3), dz)
                C0 = x.shape[-1]
                C1 = in1.shape[-1] if in1 is not None else 0
                plain = in1 is None and not up0
                if m.plan[li][0] == "flow":
                    dcat = None
                    if plain and want_mask(x) and C0 % 64 == 0:
                        db, acc = bias_of(x)
                        dcat = ops.conv3d_k3_cout3_dgrad_masked(dz, m._w[2 * li], x, db, accumulate=acc, x3=bool(self.bwd_x3))
                        premasked.add(id(x))
                    if dcat is None:
                        dcat = ops.conv3d_k3_cout3_dgrad(dz, m._w[2 * li], x3=bool(self.bwd_x3))
                elif (up0 and in1 is not None and id(in1) not in grads and id(x) not in grads and m.fold_upsampling
                      and ops.dgrad_upfold_supported(C0, C1, dz.shape[-1], self.bwd_x3, *dz.shape[:4])):
                    # concat layer, folded: the skip half is an ordinary (masked) dgrad over its C1 channels; the upsampled half
                    # comes straight out at LOW resolution from the 8-class x 8-tap fold (no full-resolution intermediate,
                    # no pooling pass), already multiplied by LeakyReLU'(x) with x's bias gradient
                    wk = m._w[2 * li]
                    wt_skip = m._book.get(("dgrad_skip", li, C0), ops.PACK_DGRAD, wk, C0, C1, bmode)
                    if want_mask(in1):
                        db, acc = bias_of(in1)
                        premasked.add(id(in1))
                        if (self.fuse_pool_bwd and id(in1) in pooled_from
                                and ops.dgrad_masked_pool_supported(C1, self.bwd_x3, *in1.shape[1:4])):
                            deferred[id(in1)] = (dz, wt_skip, C1, db, acc)     # runs at in1's pool record
                        else:
                            grads[id(in1)] = ops.conv3d_k3_dgrad_masked(dz, wt_skip, C1, in1, db, accumulate=acc, x3=self.bwd_x3)
                    else:
                        grads[id(in1)] = ops.conv3d_k3(dz, wt_skip, None, C1, leaky=False, out_f32=True, x3=self.bwd_x3)
                    wt_up = m._book.get(("dgfold", li, C0), ops.PACK_DGFOLD, wk, 0, C0, bmode)
                    if want_mask(x):
                        db, acc = bias_of(x)
                        grads[id(x)] = ops.conv3d_k3_dgrad_upfold(dz, wt_up, C0, ymask=x, dbias=db, accumulate=acc, x3=self.bwd_x3)
                        premasked.add(id(x))
                    else:
                        grads[id(x)] = ops.conv3d_k3_dgrad_upfold(dz, wt_up, C0, x3=self.bwd_x3)
                    del dz, dy
                    continue
                else:
                    wt = m._book.get(("dgrad", li), ops.PACK_DGRAD, m._w[2 * li], 0, C0 + C1, bmode)
                    if plain and want_mask(x):
                        db, acc = bias_of(x)
                        dcat = ops.conv3d_k3_dgrad_masked(dz, wt, C0, x, db, accumulate=acc, x3=self.bwd_x3)
                        premasked.add(id(x))
                    elif (up0 and in1 is not None and id(in1) not in grads
                          and ops.dgrad_split_supported(C0, C1, self.bwd_x3)):
                        # concat layer: the skip half of the gradient leaves the conv epilogue already masked, the
                        # upsampled half compact; the concatenated gradient is never written
                        kw1 = {}
                        if want_mask(in1):
                            kw1["y1"] = in1
                            kw1["dbias1"], kw1["accumulate"] = bias_of(in1)
                            premasked.add(id(in1))
                        d0c, d1 = ops.conv3d_k3_dgrad_split(dz, wt, C0, C1, x3=self.bwd_x3, **kw1)
                        grads[id(in1)] = d1
                        kw0 = {}
                        if want_mask(x):
                            kw0["y0"] = x
                            kw0["dbias0"], kw0["acc_b0"] = bias_of(x)
                            premasked.add(id(x))
                        add_grad(x, ops.upcat_bwd(d0c, C0, 0, True, **kw0)[0])
                        del dz, dy, d0c
                        continue
                    else:
                        dcat = ops.conv3d_k3(dz, wt, None, C0 + C1, leaky=False, out_f32=True, x3=self.bwd_x3)
                del dz, dy
                if plain:
                    add_grad(x, dcat)
                else:
                    kw = {}
                    if want_mask(x) and (C1 == 0 or C1 % 4 == 0):
                        kw["y0"] = x
                        kw["dbias0"], kw["acc_b0"] = bias_of(x)
                        premasked.add(id(x))
                    if in1 is not None and want_mask(in1) and C0 % 4 == 0:
                        kw["y1"] = in1
                        kw["dbias1"], kw["acc_b1"] = bias_of(in1)
                        premasked.add(id(in1))
                    d0, d1 = ops.upcat_bwd(dcat, C0, C1, up0, d_in1=grads.get(id(in1)) if in1 is not None else None, **kw)
                    add_grad(x, d0)
                    if in1 is not None:
                        grads[id(in1)] = d1
                    del dcat
            elif kind == "pool":
                _, x, p = rec
                dp = grads.pop(id(p))
                if id(x) in deferred:
                    dzd, wtd, Cd, db, acc = deferred.pop(id(x))
                    grads[id(x)] = ops.conv3d_k3_dgrad_masked(dzd, wtd, Cd, x, db, accumulate=acc, x3=self.bwd_x3, pool_grad=dp)
                elif want_mask(x):
                    db, acc = bias_of(x)
                    grads[id(x)] = ops.maxpool3d2_bwd(x, dp, dx=grads.get(id(x)), masked=True, dbias=db, acc_b=acc)
                    premasked.add(id(x))
                else:
                    grads[id(x)] = ops.maxpool3d2_bwd(x, dp, dx=grads.get(id(x)))
            elif kind == "conv0":
                _, li, src, trg, y = rec
                dy = grads.pop(id(y))
                dz = dy if id(y) in premasked else ops.leaky_bwd_bias_(y, dy, self.g[1], leaky=True)
                wgrad(lambda: ops.conv3d_k3_cin2_wgrad(src, trg, dz, self.g[0], x3=bool(self.bwd_x3)), dz)
        if deferred:   # a deferred skip gradient whose pooling record never came: the tape is not a U-Net's
            raise RuntimeError(f"{len(deferred)} deferred skip gradient(s) were never completed (tape without the pooling record)")
        if side is not None:   # join: Adam / the all-reduce read every weight gradient
            done = torch.cuda.Event()
            done.record(side)
            torch.cuda.current_stream().wait_event(done)
        if enc_ev0 is not None:
            enc_ev1 = torch.cuda.Event(enable_timing=True)
            enc_ev1.record()
            self.encoder_bwd_events.append((enc_ev0, enc_ev1))
        return grads

    # ------------------------------------------------------------------ one step
    def _render(self, src_labels, trg_labels, draws_1=None, draws_2=None):
        g1 = self.gen_1.generate(src_labels, draws=draws_1, want_onehot=False)
        g2 = self.gen_2.generate(trg_labels, draws=draws_2, want_onehot=False)
        return g1["image"], g2["image"], g1["labels"], g2["labels"]

    def _render_ahead(self, labels, ready):
        """The two generator renderings of the NEXT step's label maps on a stream of their own, queued after this step's
        kernels: ~60 launches of small HBM- / latency-bound kernels (Perlin fields, five compose steps at half resolution,
        blurs, min-max) that leave most of the chip idle when they run alone and fit beside the matrix-core kernels.  Keras
        does the analogous thing on the host (``fit`` pulls the generator from a prefetch queue).  Same kernels, same host
        draws in the same order, same values; ``ready`` = main-stream event after which the label maps may be read."""
        src, trg = labels
        main = torch.cuda.current_stream()
        if self.gstream is None:
            self.gstream = torch.cuda.Stream(device=self.model._flat.device)
        self.gstream.wait_event(ready)
        with torch.cuda.stream(self.gstream):
            out = self._render(src, trg)
            done = torch.cuda.Event()
            done.record(self.gstream)
        for t in (src, trg):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self.gstream)
        for t in out:
            t.record_stream(main)
        self._ahead = (src, trg, out, done)

    def forward_backward(self, src_labels, trg_labels, draws_1=None, draws_2=None, train=True, next_labels=None):
        """src/trg label maps uint8 [b,*S,1] of this rank's shard. Fills self.gflat (unreduced)."""
        if self._ahead is not None and train and draws_1 is None and draws_2 is None:
            a_src, a_trg, rendered, done = self._ahead
            self._ahead = None
            if a_src is not src_labels or a_trg is not trg_labels:
                raise RuntimeError("train_step(next_labels=) announced other label maps than the ones this step was called with")
            torch.cuda.current_stream().wait_event(done)
            ima_1, ima_2, lab1, lab2 = rendered
        else:
            ima_1, ima_2, lab1, lab2 = self._render(src_labels, trg_labels, draws_1, draws_2)
        b = ima_1.shape[0]
        tape = []
        flow = self._forward(ima_1, ima_2, tape)
        if next_labels is not None and train and self.render_at == "tail":
            # the next pair's renderings start when the flow head has finished: ~1.3 ms of small tail kernels (resize, VecInt,
            # warp, the losses and their adjoints) follow that leave the chip as idle as the generator's own kernels do
            after = torch.cuda.Event()
            after.record(torch.cuda.current_stream())
            self._render_ahead(next_labels, after)
        svf, steps, pos_lo, pos = self._tail_forward(flow)
        dice, top_bot = ops.dice_labels_fwd(lab1, lab2, pos, self.L, zeropad=self.zero_pad_dice)
        gl = ops.grad_l2_loss(pos, self.reg_param)
        out = {"dice": dice, "grad": gl, "loss": (dice + 1.0) + gl.mean(), "pos_flow": pos, "preint_flow": svf}
        if not train:
            return out
        # d/dflow of sum_b [(dice + 1) + grad_b]   (Keras sums the per-replica loss vector)
        dpos = ops.dice_labels_bwd(lab1, lab2, pos, top_bot, self.L, scale=float(b), zeropad=self.zero_pad_dice)
        ops.grad_l2_bwd(pos, self.reg_param, scale=1.0, out=dpos)
        dflow = self._tail_backward(dpos, svf, steps)
        self._backward(tape, dflow)
        return out

    def train_step(self, src_labels, trg_labels, draws_1=None, draws_2=None, next_labels=None):
        """next_labels=(src, trg): the label maps the NEXT ``train_step`` will be called with (the same objects); their two
        renderings are queued on the generator stream behind this step's kernels (``_render_ahead``)."""
        if next_labels is not None and not self.model._flat.is_cuda:
            next_labels = None
        if next_labels is not None and self.render_at != "tail":
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())   # whatever produced next_labels is already queued
        if self._ar_early is not None:   # a previous step died between issuing bucket 1 and waiting for it
            raise RuntimeError("train_step: an asynchronous gradient all-reduce of an earlier step is still pending")
        self._reduce_early = self.early_reduce
        try:
            out = self.forward_backward(src_labels, trg_labels, draws_1, draws_2, train=True, next_labels=next_labels)
        except BaseException:
            # bucket 1 may already be in flight on every rank: retire it before the error leaves this step, so that a caller
            # who catches it does not issue the next collective against a stale work object
            h, self._ar_early = self._ar_early, None
            if h is not None:
                try:
                    h.wait()
                except Exception:
                    pass
            raise
        finally:
            self._reduce_early = False
        if self.world > 1 or parallel.forced():
            n_late = self.goff[2 * (self._bucket_li + 1)] if self._ar_early is not None else self.gflat.numel()
            # the EXPOSED part of the exchange: with two buckets, the encoder's few hundred KB plus the wait for bucket 1
            # (which has been running under the encoder's backward); tagged with the bytes reduced inside this region
            with ops._Timed("comm:allreduce_grads", (self.world, int(self._ar_early is not None)), float(n_late * 4)):
                if self._ar_early is not None:   # bucket 1 (5.4 of 5.8 MB at 64 f) has been running since the last encoder conv
                    parallel.allreduce_sum_(self.gflat[:n_late], self.pg)
                    self._ar_early.wait()
                    self._ar_early = None
                else:
                    parallel.allreduce_sum_(self.gflat, self.pg)
        self.opt.apply(self.model._flat, self.gflat, grad_scale=1.0 / self.world)
        if self.batch_repack:
            self.model.repack()   # all weight images of the next step in one launch
        else:
            self.model.invalidate_packed()
        if next_labels is not None and self.render_at != "tail":
            self._render_ahead(next_labels, ready)
        return out

    def test_step(self, src_labels, trg_labels):
        return self.forward_backward(src_labels, trg_labels, train=False)

    # ------------------------------------------------------------------ fit
    def fit(self, gen, validation_data=None, validation_steps=0, initial_epoch=0, epochs=1, steps_per_epoch=1,
            save_name=None, save_freq=0, verbose=1, log=None, local_batches=False, local_val_batches=None, render_ahead=True):
        """Keras-``fit``-like loop over a ``gen_synthmorph_eb`` generator.

        local_batches=False: ``gen`` yields the GLOBAL batch and this rank takes rows rank*b:(rank+1)*b (what
        MirroredStrategy does with the reference's generator, train_synthmorph.py:193-194,335-344).
        local_batches=True: ``gen`` already yields only this rank's rows (``run_training`` builds it that way, so no
        rank draws, flips or masks volumes it then throws away).  The same for ``validation_data`` through
        ``local_val_batches`` (default: like ``local_batches``); a validation batch that is drawn globally but is not
        a multiple of the world size is evaluated whole on every rank.  The logged losses are means over all ranks.
        render_ahead: inside an epoch the next batch is pulled from ``gen`` one step early and its two renderings run on the
        generator stream beside the current step (``train_step(next_labels=)``); every random stream is still consumed in the
        same order, so the run is the same run.  Nothing is pulled across an epoch end (validation, checkpoint)."""
        if local_val_batches is None:
            local_val_batches = local_batches
        dev = self.model._flat.device
        history = []
        for epoch in range(initial_epoch, epochs):
            t0 = time.perf_counter()
            losses = []
            def pull():
                (src, trg), _void = next(gen)
                if not local_batches:
                    sl = parallel.shard_rows(src.shape[0], self.rank, self.world)
                    src, trg = src[sl], trg[sl]
                return src, trg
            nxt = None
            for i in range(steps_per_epoch):
                cur = nxt if nxt is not None else pull()
                nxt = pull() if (render_ahead and dev.type == "cuda" and i + 1 < steps_per_epoch) else None
                o = self.train_step(*cur, next_labels=nxt)
                losses.append(o["loss"])
            mean_loss = parallel.allreduce_mean_scalar(float(torch.stack(losses).mean()), dev, self.pg)
            rec = {"epoch": epoch + 1, "loss": mean_loss, "s_per_step": (time.perf_counter() - t0) / steps_per_epoch}
            if validation_data is not None and validation_steps:
                vl, reduce_val = [], True
                for _ in range(validation_steps):
                    (src, trg), _void = next(validation_data)
                    if not local_val_batches:
                        if src.shape[0] % self.world == 0:
                            sl = parallel.shard_rows(src.shape[0], self.rank, self.world)
                            src, trg = src[sl], trg[sl]
                        else:
                            reduce_val = False   # identical whole batch on every rank
                    vl.append(self.test_step(src, trg)["loss"])
                v = float(torch.stack(vl).mean())
                rec["val_loss"] = parallel.allreduce_mean_scalar(v, dev, self.pg) if reduce_val else v
            history.append(rec)
            if verbose and self.rank == 0:
                print(f"Epoch {epoch + 1}/{epochs} - loss: {rec['loss']:.4f}"
                      + (f" - val_loss: {rec['val_loss']:.4f}" if "val_loss" in rec else "")
                      + f" - {rec['s_per_step'] * 1e3:.0f} ms/step", flush=True)
            if log is not None and self.rank == 0:
                log(rec)
            if save_name and save_freq and (epoch + 1) % save_freq == 0 and self.rank == 0:
                self.model.save(save_name.format(epoch=epoch + 1))
        return history


def run_training(config, label_maps=None, device="cuda", rank=0, world_size=1, process_group=None, seed=0,
                 compute_dtype="fp32x3", checkpoint_ext=".h5"):
    """``train_synthmorph.py __main__`` driven by the reference's 44-key JSON config (config/config.json).
    Checkpoints are ``{epoch:04d}.h5`` in the Keras layout like the reference's ModelCheckpoint (:313-317);
    ``checkpoint_ext='.safetensors'`` selects the native format.  ``compute_dtype``: 'fp32x3' (default, as
    tools/train.py and bench.py: fp32-grade products on the bf16 matrix cores) or 'fp32' (exact fp32 MFMA).

    Data-parallel layout (one process per GPU): label maps are synthesised shard-wise (map i on rank
    i % world) and exchanged once; every rank then feeds ITS rows of the global batch from its own generator
    (batch_size // world rows, rank-dependent seed); only rank 0 writes files."""
    from . import data, networks, py_utils, synth
    data_cfg = config
    if data_cfg["batch_size"] % world_size:
        raise ValueError(f"batch_size {data_cfg['batch_size']} is not a multiple of the number of GPUs {world_size} "
                         "(train_synthmorph.py:193-194)")
    if label_maps is None:
        if data_cfg["gen_label"]:
            mine = synth.generate_label_maps(data_cfg["in_shape"], data_cfg["num_labels"], data_cfg["num_maps"],
                                             data_cfg["im_scales"], data_cfg["def_scales"], data_cfg["im_max_std"],
                                             data_cfg["def_max_std"], seed=seed, device=device,
                                             shard=(rank, world_size))
            label_maps = parallel.gather_maps(mine, data_cfg["num_maps"], rank, world_size, device, process_group)
            if data_cfg["save_label"] and rank == 0:  # label_map_{add_str}{i}.nii.gz, train_synthmorph.py:71-76
                os.makedirs(data_cfg["label_dir"], exist_ok=True)
                for i, m in enumerate(label_maps):
                    py_utils.write_nifti(m, os.path.join(data_cfg["label_dir"], f"label_map_{data_cfg['add_str']}{i + 1}.nii.gz"),
                                         np.eye(4))
        else:  # vxm.py.utils.load_labels (train_synthmorph.py:207): .nii.gz / .nii / .npz / .npy, clear error when empty
            _, label_maps = py_utils.load_labels(data_cfg["label_dir"])
    labels_in = np.unique(label_maps)
    np.random.seed(42)  # train_synthmorph.py:209
    label_maps = list(label_maps)
    np.random.shuffle(label_maps)
    n_tr = int(len(label_maps) * data_cfg["train_frac"])
    maps_tr, maps_val = label_maps[:n_tr], label_maps[n_tr:]
    if data_cfg["gen_label_only"]:
        return None
    b_loc = data_cfg["batch_size"] // world_size
    val_sharded = data_cfg["batch_size_val"] % world_size == 0
    bv_loc = data_cfg["batch_size_val"] // world_size if val_sharded else data_cfg["batch_size_val"]
    gen_tr = data.gen_synthmorph_eb(maps_tr, batch_size=b_loc, same_subj=data_cfg["same_subj"], flip=True,
                                    random_zero_borders=data_cfg["zero_borders_maps"],
                                    scale_zero_borders=data_cfg["zero_bord_scale"], frac_zero_bord=data_cfg["zero_bord_frac"],
                                    rng=np.random.default_rng([seed, rank]), device=device)  # label maps stay in HBM
    gen_val = data.gen_synthmorph_eb(maps_val, batch_size=bv_loc, same_subj=data_cfg["same_subj"],
                                     flip=True, random_zero_borders=data_cfg["zero_borders_maps_val"],
                                     scale_zero_borders=data_cfg["zero_bord_scale"],
                                     frac_zero_bord=data_cfg["zero_bord_frac"],
                                     rng=np.random.default_rng([seed + 1, rank if val_sharded else 0]),
                                     device=device) if maps_val else None
    in_shape = label_maps[0].shape
    gen_args = dict(in_shape=in_shape, in_label_list=labels_in, out_label_list=labels_in, warp_std=data_cfg["vel_std"],
                    warp_res=data_cfg["vel_res"], blur_std=data_cfg["blur_std"], bias_std=data_cfg["bias_std"],
                    bias_res=data_cfg["bias_res"], gamma_std=data_cfg["gamma"], device=device)
    g1 = synth.labels_to_image(**gen_args, id=0, seed=seed * 2 + 11 + rank * 1000)
    g2 = synth.labels_to_image(**gen_args, id=1, seed=seed * 2 + 12 + rank * 1000)
    model = networks.VxmDense(in_shape, int_steps=data_cfg["int_steps"], int_resolution=data_cfg["int_res"],
                              svf_resolution=data_cfg["svf_res"], nb_unet_features=(data_cfg["enc"], data_cfg["dec"]),
                              compute_dtype=compute_dtype, device=device, seed=seed)
    if data_cfg["bool_init_weights"]:
        model.load_weights(data_cfg["init_weights"])
    model_dir = os.path.join(data_cfg["model_dir"], data_cfg["sub_dir"]) if data_cfg["bool_sub_dir"] else data_cfg["model_dir"]
    if rank == 0:
        os.makedirs(model_dir, exist_ok=True)
    save_name = os.path.join(model_dir, "{epoch:04d}" + checkpoint_ext)
    trainer = SynthMorphTrainer(model, g1, g2, reg_param=data_cfg["reg_param"], optimizer=Adam(data_cfg["lr"]),
                                zero_pad_dice=data_cfg["zero_borders_maps"] or data_cfg["zero_borders_maps_val"],
                                process_group=process_group, world_size=world_size, rank=rank)
    if rank == 0:
        model.save(save_name.format(epoch=data_cfg["init_epoch"]))
    steps = max(len(maps_tr) // data_cfg["batch_size"], 1)
    hist = trainer.fit(gen_tr, validation_data=gen_val,
                       validation_steps=(len(maps_val) // data_cfg["batch_size_val"]) if maps_val else 0,
                       initial_epoch=data_cfg["init_epoch"], epochs=data_cfg["epochs"], steps_per_epoch=steps,
                       save_name=save_name, save_freq=data_cfg["save_freq"], verbose=data_cfg["verbose"],
                       local_batches=True, local_val_batches=True)
    return trainer, hist
