"""Same-box A/B of single kernels: this tree's libmmr_hip.so against round 4's (tools/ab/libmmr_hip_r04.so, built by
tools/ab/build_r04_lib.sh from git 09a1874), both loaded into ONE process through raw ctypes and timed alternately
(3 rounds of 20 launches each, HIP events) on the same tensors.  Boxes of the pool differ by 5 - 15 %; only alternated runs on one
box rank two builds.

    python tools/ab_kernels.py [upfold] [conv] [flow] [thin] [wgrad] [compose] [ncc] [nccbwd] [bending] [cin2]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mmr
from mmr import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
new = _lib.load()
old = ctypes.CDLL(os.path.join(ROOT, "tools", "ab", "libmmr_hip_r04.so"))
for name, (res, args) in _lib.SIGNATURES.items():
    if hasattr(old, name):
        getattr(old, name).restype = res
        getattr(old, name).argtypes = args
def _bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = args
    return lib


# extra builds of the CURRENT sources with experiment flags (tools/ab/build_variant.sh): every tools/ab/libmmr_hip_<name>.so
VARIANTS = {os.path.basename(p)[len("libmmr_hip_"):-3]: _bind(os.path.join(ROOT, "tools", "ab", p))
            for p in sorted(os.listdir(os.path.join(ROOT, "tools", "ab"))) if p.startswith("libmmr_hip_") and p.endswith(".so") and "r04" not in p}
dev = torch.device("cuda", 0)
st = lambda: torch.cuda.current_stream().cuda_stream
which = set(sys.argv[1:]) or {"flow", "thin", "ncc", "bending", "cin2", "upfold"}


def ab(label, f_old, f_new, rounds=3, n=20, check=None):
    res = {"old": [], "new": []}
    for fn in (f_old, f_new):
        assert fn() == 0, label
    torch.cuda.synchronize()
    if check is not None:
        check()
    for _ in range(rounds):
        for key, fn in (("old", f_old), ("new", f_new)):
            for _ in range(3):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize()
            res[key].append(a.elapsed_time(b) / n * 1e3)
    o, w = min(res["old"]), min(res["new"])
    print(f"{label:58s} r04 {o:8.1f} us   now {w:8.1f} us   {100 * (w - o) / o:+6.1f} %   (all: "
          f"{', '.join(f'{x:.1f}' for x in res['old'])} | {', '.join(f'{x:.1f}' for x in res['new'])})", flush=True)


g = torch.Generator(device="cpu").manual_seed(0)
if "upfold" in which:
    # launch A of dec_final_0 at C2: the upsampled half of concat([up2(256 ch @ 80x80x96), skip]) -> 256 on the low-resolution grid,
    # 8 parity classes x 8 taps, IEEE-half partial at 160x160x192 (mmr_conv3d_k3_upfold_fwd)
    for (S2, C0, Cout, dt, half, tag) in (((80, 80, 96), 256, 256, 1, 1, "bf16 256 -> 256, partial at 160x160x192 (C2 dec_final_0)"),
                                          ((40, 40, 48), 256, 256, 1, 1, "bf16 256 -> 256, partial at 80x80x96 (C2 dec_conv_3)"),
                                          ((80, 80, 80), 64, 64, 2, 0, "fp32x3 64 -> 64, partial at 160^3 (C3 dec_final_0)")):
        xl = (torch.randn((1,) + S2 + (C0,), generator=g) * 0.5).to(torch.bfloat16 if dt == 1 else torch.float32).to(dev)
        wk = (torch.randn((27, C0, Cout), generator=g) * 0.02).to(dev)
        nb = int(new.mmr_conv3d_k3_upfold_packed_bytes(C0, Cout, dt))
        wp = torch.empty(nb, dtype=torch.uint8, device=dev)
        assert new.mmr_conv3d_k3_upfold_pack(wk.data_ptr(), wp.data_ptr(), C0, Cout, dt, st()) == 0
        full = tuple(2 * v for v in S2)
        p1 = torch.empty((1,) + full + (Cout,), dtype=torch.float16 if half else torch.float32, device=dev)
        p2 = torch.empty_like(p1)
        call = lambda lib, o: lib.mmr_conv3d_k3_upfold_fwd(xl.data_ptr(), C0, wp.data_ptr(), o.data_ptr(), half, 1, *S2, Cout, dt, st())

        def chk():
            assert torch.equal(p1, p2), float((p1.float() - p2.float()).abs().max())
        if hasattr(new, "mmr_conv_set_xcd_pair"):      # same library, class order per XCD off | on
            def off():
                new.mmr_conv_set_xcd_pair(0)
                return call(new, p1)

            for P in (2, 4, 8):
                def on():
                    new.mmr_conv_set_xcd_pair(P)
                    return call(new, p2)
                ab("upfold " + tag + f" (blockIdx.y = class | {P} classes per group of {P} XCDs)", off, on, check=chk, n=10)
            new.mmr_conv_set_xcd_pair(2)
        ab("upfold " + tag, lambda: call(old, p1), lambda: call(new, p2), check=chk, n=10)
        for vn, vl in VARIANTS.items():
            if "uploop" in vn:
                ab(f"  ... variant {vn}", lambda: call(old, p1), lambda: call(vl, p2), check=chk, n=10)
            if "skip" in vn:    # timing-only builds (wrong results): bounds on what fewer A restages could buy
                ab(f"  ... TIMING-ONLY {vn}", lambda: call(old, p1), lambda: call(vl, p2), check=None, n=10)
        del xl, p1, p2
if "conv" in which:
    # the plain 27-tap bf16 256 -> 256 conv of C2's top level (enc / dec layers at 160x160x192 are this kernel or its CINIT form)
    S, C = (160, 160, 192), 256
    x = (torch.randn((1,) + S + (C,), generator=g) * 0.5).to(torch.bfloat16).to(dev)
    wk = (torch.randn((27, C, C), generator=g) * 0.02).to(dev)
    bias = torch.zeros(C, device=dev)
    wp = torch.empty(int(new.mmr_conv3d_k3_packed_bytes(C, C, 1)), dtype=torch.uint8, device=dev)
    assert new.mmr_conv3d_k3_pack(wk.data_ptr(), wp.data_ptr(), C, C, 1, 0, st()) == 0
    o1, o2 = torch.empty((1,) + S + (C,), dtype=torch.bfloat16, device=dev), torch.empty((1,) + S + (C,), dtype=torch.bfloat16, device=dev)
    call = lambda lib, o: lib.mmr_conv3d_k3_fwd(x.data_ptr(), C, 0, None, 0, wp.data_ptr(), bias.data_ptr(), o.data_ptr(), None, 1, *S, C, 1, 0.2, 1, 0, st())

    def chk():
        assert torch.equal(o1, o2)
    ab("conv bf16 256 -> 256, 160x160x192 (C2 top level)", lambda: call(old, o1), lambda: call(new, o2), check=chk, n=5)
    for vn in ("xo1", "xo2"):
        if vn in VARIANTS:
            ab(f"  ... tile order plain | variant {vn} (XCD-contiguous runs{' in 2x4x4 blocks' if vn == 'xo2' else ''})", lambda: call(new, o1), lambda: call(VARIANTS[vn], o2), check=chk, n=5)
    del x, o1, o2
if "flow" in which:
    S, C = (160, 160, 192), 256
    x = (torch.randn((1,) + S + (C,), generator=g) * 0.5).to(torch.bfloat16).to(dev)
    w = (torch.randn((3, 3, 3, C, 3), generator=g) * 0.02).to(dev)
    b = torch.zeros(3, device=dev)
    o1, o2 = torch.empty((1,) + S + (3,), device=dev), torch.empty((1,) + S + (3,), device=dev)
    call = lambda lib, o: lib.mmr_conv3d_k3_cout3_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, *S, C, 1, st())

    def chk():
        assert torch.equal(o1, o2), float((o1 - o2).abs().max())
    ab("flow head bf16, 160x160x192 x 256 (C2)", lambda: call(old, o1), lambda: call(new, o2), check=chk)
    if "noxcd" in VARIANTS:
        ab("  ... same sources, tile walk plain (-DMMR_NO_XCD_TILES) | XCD-contiguous", lambda: call(VARIANTS["noxcd"], o1), lambda: call(new, o2), check=chk)
    for vn, vl in VARIANTS.items():
        if vn.startswith("fh_"):
            ab(f"  ... variant {vn}", lambda: call(old, o1), lambda: call(vl, o2), check=chk)
    del x
    S, C = (160, 160, 160), 64
    x = torch.randn((1,) + S + (C,), generator=g).to(dev)
    w = (torch.randn((3, 3, 3, C, 3), generator=g) * 0.05).to(dev)
    o1, o2 = torch.empty((1,) + S + (3,), device=dev), torch.empty((1,) + S + (3,), device=dev)
    call = lambda lib, o: lib.mmr_conv3d_k3_cout3_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, *S, C, 2, st())
    ab("flow head fp32x3, 160^3 x 64 (C3)", lambda: call(old, o1), lambda: call(new, o2), check=chk)
if "thin" in which:
    S = (160, 160, 160)
    x = torch.randn((1,) + S + (64,), generator=g).to(dev)
    dflow = torch.randn((1,) + S + (3,), generator=g).to(dev)
    src, trg = torch.rand((1,) + S + (1,), generator=g).to(dev), torch.rand((1,) + S + (1,), generator=g).to(dev)
    d1, d2 = torch.zeros((3, 3, 3, 64, 3), device=dev), torch.zeros((3, 3, 3, 64, 3), device=dev)
    ws = torch.empty(int(new.mmr_conv3d_k3_wgrad_ws_bytes(1, *S, 64, 3)) + 1024, dtype=torch.uint8, device=dev)
    call = lambda lib, d: lib.mmr_conv3d_k3_wgrad_f32x3(x.data_ptr(), 64, 0, None, 0, dflow.data_ptr(), d.data_ptr(), ws.data_ptr(), 1, *S, 3, 0, st())

    def chk():
        assert float((d1 - d2).abs().max() / d1.abs().max()) < 1e-6
    ab("flow-head weight gradient fp32x3 (thin_wgrad_x3), 160^3 x 64", lambda: call(old, d1), lambda: call(new, d2), check=chk)
    e1, e2 = torch.zeros((3, 3, 3, 2, 64), device=dev), torch.zeros((3, 3, 3, 2, 64), device=dev)
    ws2 = torch.empty(int(new.mmr_conv3d_k3_cin2_wgrad_ws_bytes(64)) + 1024, dtype=torch.uint8, device=dev)
    call = lambda lib, d: lib.mmr_conv3d_k3_cin2_wgrad_f32x3(src.data_ptr(), trg.data_ptr(), x.data_ptr(), d.data_ptr(), ws2.data_ptr(), 1, *S, 64, 0, st())

    def chk2():
        assert float((e1 - e2).abs().max() / e1.abs().max()) < 1e-6
    ab("first-layer weight gradient fp32x3 (thin_wgrad_x3), 160^3 x 64", lambda: call(old, e1), lambda: call(new, e2), check=chk2)
    wfh = (torch.randn((3, 3, 3, 64, 3), generator=g) * 0.05).to(dev)
    dx1, dx2 = torch.empty_like(x), torch.empty_like(x)
    db1, db2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    ws3 = torch.empty(int(new.mmr_conv3d_k3_cout3_dgrad_masked_ws_bytes(1, *S, 64)) + 1024, dtype=torch.uint8, device=dev)
    calld = lambda lib, dx, db: lib.mmr_conv3d_k3_cout3_dgrad_masked_f32x3(dflow.data_ptr(), wfh.data_ptr(), dx.data_ptr(), 1, *S, 64, x.data_ptr(), 0.2, db.data_ptr(), ws3.data_ptr(), 0, st())

    def chk3():
        assert torch.equal(dx1, dx2) and float((db1 - db2).abs().max() / db1.abs().max()) < 1e-5
    ab("flow-head data gradient + mask + bias sums fp32x3 (flow_dgrad_x3), 160^3 x 64", lambda: calld(old, dx1, db1), lambda: calld(new, dx2, db2), check=chk3)
    if "noxcd" in VARIANTS:
        nx = VARIANTS["noxcd"]
        call = lambda lib, d: lib.mmr_conv3d_k3_wgrad_f32x3(x.data_ptr(), 64, 0, None, 0, dflow.data_ptr(), d.data_ptr(), ws.data_ptr(), 1, *S, 3, 0, st())
        ab("  ... thin_wgrad_x3: tile walk plain (-DMMR_NO_XCD_TILES) | XCD-contiguous", lambda: call(nx, d1), lambda: call(new, d2), check=chk)
        ab("  ... flow_dgrad_x3: tile walk plain | XCD-contiguous", lambda: calld(nx, dx1, db1), lambda: calld(new, dx2, db2), check=chk3)
if "wgrad" in which:
    S = (160, 160, 160)
    x = torch.randn((1,) + S + (64,), generator=g).to(dev)
    dz = torch.randn((1,) + S + (64,), generator=g).to(dev)
    d1, d2 = torch.zeros((3, 3, 3, 64, 64), device=dev), torch.zeros((3, 3, 3, 64, 64), device=dev)
    ws = torch.empty(int(new.mmr_conv3d_k3_wgrad_ws_bytes(1, *S, 64, 64)) + 1024, dtype=torch.uint8, device=dev)
    call = lambda lib, d: lib.mmr_conv3d_k3_wgrad_f32x3(x.data_ptr(), 64, 0, None, 0, dz.data_ptr(), d.data_ptr(), ws.data_ptr(), 1, *S, 64, 0, st())

    def chk():
        assert float((d1 - d2).abs().max() / d1.abs().max()) < 1e-5
    ab("weight gradient fp32x3 64 -> 64, 160^3 (wgrad_x3_kernel)", lambda: call(old, d1), lambda: call(new, d2), check=chk, n=10)
    if "noxcd" in VARIANTS:
        ab("  ... tile walk plain (-DMMR_NO_XCD_TILES) | XCD-contiguous", lambda: call(VARIANTS["noxcd"], d1), lambda: call(new, d2), check=chk, n=10)
    del x, dz
if "compose" in which:
    for S in ((80, 80, 80), (80, 80, 96), (160, 160, 192)):
        vel = (torch.randn((1,) + S + (3,), generator=g) * 3).to(dev)
        o1, o2, tmp = torch.empty_like(vel), torch.empty_like(vel), torch.empty_like(vel)
        f_old = lambda: old.mmr_vecint_f32(vel.data_ptr(), o1.data_ptr(), tmp.data_ptr(), 1, *S, 5, st())
        f_new = lambda: new.mmr_vecint_f32(vel.data_ptr(), o2.data_ptr(), tmp.data_ptr(), 1, *S, 5, st())

        def chk():
            assert torch.equal(o1, o2), float((o1 - o2).abs().max())
        ab(f"VecInt, 5 squaring steps, {S} (thread per voxel x channel | per voxel), bit-identical", f_old, f_new, check=chk)
if "ncc" in which or "bending" in which or "nccbwd" in which:
    S = (256, 256, 256)
    I, J = torch.rand((1,) + S + (1,), generator=g).to(dev), torch.rand((1,) + S + (1,), generator=g).to(dev)
    flow = torch.randn((1,) + S + (3,), generator=g).to(dev)
    o1, o2 = torch.empty(1, device=dev), torch.empty(1, device=dev)
    ticket = torch.zeros(4, dtype=torch.int32, device=dev)
    if "ncc" in which:
        ws = torch.empty(int(new.mmr_ncc_ws_bytes(1, *S)) + 1024, dtype=torch.uint8, device=dev)
        f_old = lambda: old.mmr_ncc_fwd_f32(I.data_ptr(), J.data_ptr(), o1.data_ptr(), ws.data_ptr(), 1, *S, 9, 1e-5, 0, st())
        f_new = lambda: new.mmr_ncc_fwd_ticket_f32(I.data_ptr(), J.data_ptr(), o2.data_ptr(), ws.data_ptr(), ticket.data_ptr(), 1, *S, 9, 1e-5, 0, 1.0, 0, st())

        def chk():
            assert abs(float(o1) - float(o2)) < 1e-6 * abs(float(o1)), (float(o1), float(o2))
        ab("NCC(9) forward 256^3 (kernel + finalize | in-kernel finalize)", f_old, f_new, check=chk)
        f_new2 = lambda: new.mmr_ncc_fwd_f32(I.data_ptr(), J.data_ptr(), o2.data_ptr(), ws.data_ptr(), 1, *S, 9, 1e-5, 0, st())
        ab("NCC(9) forward 256^3 (both with the finalize launch)", f_old, f_new2, check=chk)
    if "nccbwd" in which:
        wso = torch.empty(int(old.mmr_ncc_bwd_ws_bytes(1, *S)) + 1024, dtype=torch.uint8, device=dev)
        wsn = torch.empty(int(new.mmr_ncc_bwd_ws_bytes(1, *S)) + 1024, dtype=torch.uint8, device=dev)
        a1, b1, a2, b2 = (torch.empty_like(I) for _ in range(4))
        f_old = lambda: old.mmr_ncc_bwd_f32(I.data_ptr(), J.data_ptr(), None, a1.data_ptr(), b1.data_ptr(), wso.data_ptr(), 1, *S, 9, 1e-5, 0, st())
        f_new = lambda: new.mmr_ncc_bwd_f32(I.data_ptr(), J.data_ptr(), None, a2.data_ptr(), b2.data_ptr(), wsn.data_ptr(), 1, *S, 9, 1e-5, 0, st())

        def chk():
            e = max(float((a1 - a2).abs().max() / a1.abs().max()), float((b1 - b2).abs().max() / b1.abs().max()))
            assert e < 2e-5, e
        if hasattr(new, "mmr_ncc_bwd_set_form"):
            for nf in (5, 3):
                new.mmr_ncc_bwd_set_form(nf)
                ab(f"NCC(9) BACKWARD 256^3, dI + dJ (four separable launches | coefficients + box filter, NF = {nf})", f_old, f_new, check=chk, n=5)
            new.mmr_ncc_bwd_set_form(-1)
        else:
            ab("NCC(9) BACKWARD 256^3, dI + dJ (four separable launches | coefficients + box filter)", f_old, f_new, check=chk, n=5)
        f_old1 = lambda: old.mmr_ncc_bwd_f32(I.data_ptr(), J.data_ptr(), None, None, b1.data_ptr(), wso.data_ptr(), 1, *S, 9, 1e-5, 0, st())
        f_new1 = lambda: new.mmr_ncc_bwd_f32(I.data_ptr(), J.data_ptr(), None, None, b2.data_ptr(), wsn.data_ptr(), 1, *S, 9, 1e-5, 0, st())

        def chk1():
            e = float((b1 - b2).abs().max() / b1.abs().max())
            assert e < 2e-5, e
        ab("NCC(9) BACKWARD 256^3, dJ only", f_old1, f_new1, check=chk1, n=5)
    if "bending" in which:
        ws = torch.empty(int(new.mmr_bending_ws_bytes(1, *S)) + 1024, dtype=torch.uint8, device=dev)
        f_old = lambda: old.mmr_bending_fwd_f32(flow.data_ptr(), o1.data_ptr(), ws.data_ptr(), 1, *S, st())
        f_new = lambda: new.mmr_bending_fwd_ticket_f32(flow.data_ptr(), o2.data_ptr(), ws.data_ptr(), ticket.data_ptr(), 1, *S, 1.0, 0, st())

        def chk():
            assert float(o1) == float(o2), (float(o1), float(o2))
        ab("bending energy forward 256^3 (kernel + finalize | in-kernel)", f_old, f_new, check=chk)
        g1, g2 = torch.empty_like(flow), torch.empty_like(flow)
        f_old = lambda: old.mmr_bending_bwd_f32(flow.data_ptr(), None, g1.data_ptr(), 1, *S, 0, st())
        f_new = lambda: new.mmr_bending_bwd_f32(flow.data_ptr(), None, g2.data_ptr(), 1, *S, 0, st())

        def chk():
            assert float((g1 - g2).abs().max() / g1.abs().max()) < 1e-5, float((g1 - g2).abs().max() / g1.abs().max())
        ab("bending energy BACKWARD 256^3 (gather | tiled)", f_old, f_new, check=chk, n=5)
        for vn, vl in VARIANTS.items():
            if vn.startswith("bb"):
                ab(f"  ... TIMING-ONLY {vn}", f_old, lambda: vl.mmr_bending_bwd_f32(flow.data_ptr(), None, g2.data_ptr(), 1, *S, 0, st()), n=5)
