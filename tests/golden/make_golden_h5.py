#!/opt/conda/bin/python3.9
"""Writes the HDF5 fixtures that pin mmr.h5lite (run with an interpreter that has h5py:
``/opt/conda/bin/python3.9 tests/golden/make_golden_h5.py``; h5py 3.3.0 / libhdf5 1.10.6 in this image).

keras_like_vxm.h5     -- the layout Keras 2.x ``Model.save(path.h5)`` produces for a (tiny) VxmDense: root attributes
                         keras_version / backend / model_config, group ``model_weights`` with ``layer_names`` and one
                         group per layer holding ``weight_names`` + the kernel / bias datasets, an ``optimizer_weights``
                         group (layout restated from keras/saving/hdf5_format.py of Keras 2.7; the arrays are seeded
                         random numbers, no reference weights exist offline).
keras_like_weights.h5 -- ``Model.save_weights(path.h5)`` layout (layer groups at the root).
h5_variants.h5        -- the same arrays stored chunked / gzip / shuffle, compact, big-endian, with variable-length
                         string attributes and a 40-member group (multi-node group B-tree).
keras_like_expected.npz -- the arrays, for the reader test.
"""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ENC, DEC = [4, 6], [6, 4, 4]


def plan():
    out, skips, cin = [], [], 2
    for i, nf in enumerate(ENC):
        out.append((f"unet_enc_conv_{i}_0", cin, nf))
        skips.append(nf)
        cin = nf
    for i in range(len(ENC)):
        out.append((f"unet_dec_conv_{i}_0", cin, DEC[i]))
        cin = DEC[i] + skips.pop()
    for j, nf in enumerate(DEC[len(ENC):]):
        out.append((f"unet_dec_final_conv_{j}", cin, nf))
        cin = nf
    out.append(("flow", cin, 3))
    return out


def save_attrs(group, name, data):
    group.attrs[name] = data


def write_weights_group(g, layers, weights):
    save_attrs(g, "layer_names", np.array([n.encode("utf8") for n, _ in layers]))
    save_attrs(g, "backend", "tensorflow".encode("utf8"))
    save_attrs(g, "keras_version", "2.7.0".encode("utf8"))
    for name, wnames in layers:
        lg = g.create_group(name)
        save_attrs(lg, "weight_names", np.array([w.encode("utf8") for w in wnames]) if wnames else np.array([], dtype="S1"))
        for w in wnames:
            val = weights[w]
            d = lg.create_dataset(w, val.shape, dtype=val.dtype)
            if val.shape:
                d[:] = val
            else:
                d[()] = val


def main():
    rng = np.random.default_rng(7)
    weights, layers = {}, [("source_input", []), ("target_input", []), ("unet_input_concat", [])]
    for name, cin, cout in plan():
        k, b = f"{name}/kernel:0", f"{name}/bias:0"
        weights[k] = rng.standard_normal((3, 3, 3, cin, cout)).astype(np.float32)
        weights[b] = rng.standard_normal((cout,)).astype(np.float32)
        layers.append((name, [k, b]))
        if name != "flow":
            layers.append((name + "_activation", []))
    layers += [("vxm_dense_flow_resize", []), ("vxm_dense_flow_int", []), ("vxm_dense_diffflow", []), ("vxm_dense_transformer", [])]
    config = dict(inshape=[16, 16, 16], nb_unet_features=[ENC, DEC], nb_unet_levels=None, unet_feat_mult=1,
                  nb_unet_conv_per_level=1, int_steps=5, svf_resolution=2, int_resolution=2, int_downsize=None, bidir=False,
                  use_probs=False, src_feats=1, trg_feats=1, unet_half_res=False, input_model=None, hyp_model=None,
                  fill_value=None, reg_field="preintegrated", name="vxm_dense")
    opt = {"Adam/iter:0": np.array(1234, dtype=np.int64),
           "Adam/unet_enc_conv_0_0/kernel/m:0": rng.standard_normal((3, 3, 3, 2, 4)).astype(np.float32)}

    with h5py.File(os.path.join(HERE, "keras_like_vxm.h5"), "w") as f:
        f.attrs["keras_version"] = "2.7.0".encode("utf8")
        f.attrs["backend"] = "tensorflow".encode("utf8")
        f.attrs["model_config"] = json.dumps({"class_name": "VxmDense", "config": config}).encode("utf8")
        f.attrs["training_config"] = json.dumps({"loss": None, "optimizer_config": {"class_name": "Adam"}}).encode("utf8")
        write_weights_group(f.create_group("model_weights"), layers, weights)
        og = f.create_group("optimizer_weights")
        og.attrs["weight_names"] = np.array([n.encode("utf8") for n in opt])
        for n, v in opt.items():
            d = og.create_dataset(n, v.shape, dtype=v.dtype)
            d[()] = v

    with h5py.File(os.path.join(HERE, "keras_like_weights.h5"), "w") as f:
        write_weights_group(f, layers, weights)

    with h5py.File(os.path.join(HERE, "h5_variants.h5"), "w") as f:
        a = weights["unet_dec_conv_0_0/kernel:0"]
        f.create_dataset("chunked_gzip_shuffle", data=a, chunks=(2, 3, 3, 3, 4), compression="gzip", shuffle=True)
        f.create_dataset("chunked_plain", data=a, chunks=(3, 3, 1, 6, 5))  # ragged edge chunks
        f.create_dataset("be_f64", data=a.astype(">f8"))
        f.create_dataset("f16", data=a.astype(np.float16))
        f.create_dataset("i32", data=np.arange(-5, 7, dtype=np.int32).reshape(3, 4))
        f.create_dataset("u8", data=np.arange(200, 212, dtype=np.uint8))
        f.create_dataset("scalar", data=np.float32(2.5))
        f.create_dataset("empty", shape=(0, 3), dtype=np.float32)
        f.create_dataset("bools", data=np.array([True, False, True]))
        f.create_dataset("fixed_str", data=np.array([b"ab", b"cde"]))
        f.attrs["vlen_str"] = "variable length é"  # h5py stores str as variable-length UTF-8 (global heap)
        f.attrs["vlen_list"] = np.array(["x", "yy", "zzz"], dtype=h5py.string_dtype())
        f.attrs["float_attr"] = np.float64(0.25)
        f.attrs["int_vec"] = np.arange(5, dtype=np.int64)
        big = f.create_group("big")
        for i in range(40):
            big.create_dataset(f"member_{i:02d}", data=np.full((2,), i, dtype=np.int16))
        big.create_group("nested/deeper").attrs["tag"] = b"leaf"

    with h5py.File(os.path.join(HERE, "h5_latest.h5"), "w", libver="latest") as f:
        f.attrs["keras_version"] = b"2.7.0"
        g = f.create_group("model_weights")
        g.create_dataset("k", data=weights["flow/kernel:0"])
        g.attrs["layer_names"] = np.array([b"flow"])

    np.savez_compressed(os.path.join(HERE, "keras_like_expected.npz"), config=json.dumps(config),
                        layer_names=np.array([n for n, _ in layers]),
                        **{"w::" + k: v for k, v in weights.items()}, **{"o::" + k: v for k, v in opt.items()})
    for fn in ("keras_like_vxm.h5", "keras_like_weights.h5", "h5_variants.h5", "h5_latest.h5", "keras_like_expected.npz"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
