"""Keras .h5 weight files without h5py: mmr.h5lite + VxmDense.load / save / load_weights.

The reader is pinned by fixtures written with h5py 3.3.0 / libhdf5 1.10.6 (tests/golden/make_golden_h5.py); the writer
is checked by reading its files back with h5py where an interpreter that has it exists (skipped otherwise)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import mmr
from mmr import h5lite, networks

G = os.path.join(os.path.dirname(__file__), "golden")
EXP = np.load(os.path.join(G, "keras_like_expected.npz"))
H5PY_PYTHON = "/opt/conda/bin/python3.9"


def _have_h5py():
    if not os.path.exists(H5PY_PYTHON):
        return False
    return subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True).returncode == 0


def test_reader_keras_full_model():
    with h5lite.File(os.path.join(G, "keras_like_vxm.h5")) as f:
        assert f.attrs["keras_version"] in (b"2.7.0", "2.7.0") and f.attrs["backend"] in (b"tensorflow", "tensorflow")
        cfg = json.loads(f.attrs["model_config"])
        assert cfg["class_name"] == "VxmDense" and cfg["config"] == json.loads(str(EXP["config"]))
        mw = f["model_weights"]
        assert [n.decode() for n in mw.attrs["layer_names"]] == list(EXP["layer_names"])
        assert sorted(mw.keys()) == sorted(EXP["layer_names"])
        assert mw["source_input"].attrs["weight_names"].size == 0
        for k in EXP.files:
            if k.startswith("w::"):
                name = k[3:]
                got = mw[name.split("/")[0]][name][()]
                assert got.dtype == np.float32 and np.array_equal(got, EXP[k])
            elif k.startswith("o::"):
                assert np.array_equal(f["optimizer_weights"][k[3:]][()], EXP[k])
        assert "nope" not in mw
        with pytest.raises(KeyError):
            mw["nope"]


def test_reader_variants():
    a = EXP["w::unet_dec_conv_0_0/kernel:0"]
    with h5lite.File(os.path.join(G, "h5_variants.h5")) as v:
        for n in ("chunked_gzip_shuffle", "chunked_plain"):
            assert np.array_equal(v[n][()], a)
        assert v["be_f64"][()].dtype == np.float64 and np.array_equal(v["be_f64"][()], a.astype(np.float64))
        assert np.array_equal(v["f16"][()], a.astype(np.float16))
        assert np.array_equal(v["i32"][()], np.arange(-5, 7, dtype=np.int32).reshape(3, 4))
        assert np.array_equal(v["u8"][()], np.arange(200, 212, dtype=np.uint8))
        assert v["scalar"][()] == np.float32(2.5) and v["empty"][()].shape == (0, 3)
        assert v["bools"][()].tolist() == [True, False, True]
        assert v["fixed_str"][()].tolist() == [b"ab", b"cde"]
        assert v.attrs["vlen_str"] == "variable length é" and list(v.attrs["vlen_list"]) == ["x", "yy", "zzz"]
        assert v.attrs["float_attr"] == 0.25 and v.attrs["int_vec"].tolist() == [0, 1, 2, 3, 4]
        big = v["big"]
        assert len(big.keys()) == 41  # 40 datasets + 'nested': several symbol-table nodes under the group B-tree
        for i in range(40):
            assert big[f"member_{i:02d}"][()].tolist() == [i, i]
        assert big["nested/deeper"].attrs["tag"] == b"leaf"
    with h5lite.File(os.path.join(G, "h5_latest.h5")) as l:  # superblock v3, v2 object headers, link messages
        assert np.array_equal(l["model_weights/k"][()], EXP["w::flow/kernel:0"])
        assert l["model_weights"].attrs["layer_names"].tolist() == [b"flow"]


def test_reader_rejects_non_hdf5(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 10)
    with pytest.raises(h5lite.H5Error):
        h5lite.File(str(p))


def _tiny(seed=0):
    cfg = json.loads(str(EXP["config"]))
    return networks.VxmDense(cfg["inshape"], nb_unet_features=cfg["nb_unet_features"], int_steps=cfg["int_steps"],
                             svf_resolution=cfg["svf_resolution"], int_resolution=cfg["int_resolution"],
                             compute_dtype="fp32", device="cpu", seed=seed)


def _expected_list():
    out = []
    for n in EXP["layer_names"]:
        if f"w::{n}/kernel:0" in EXP.files:
            out += [EXP[f"w::{n}/kernel:0"], EXP[f"w::{n}/bias:0"]]
    return out


def test_vxmdense_load_keras_h5():
    m = networks.VxmDense.load(os.path.join(G, "keras_like_vxm.h5"), input_model=None, compute_dtype="fp32", device="cpu")
    assert m.inshape == (16, 16, 16) and m.enc == [4, 6] and m.dec == [6, 4, 4] and m.int_steps == 5
    for got, want in zip(m.get_weights(), _expected_list()):
        assert np.array_equal(got, want)
    m2 = _tiny(seed=3)
    m2.load_weights(os.path.join(G, "keras_like_weights.h5"))  # save_weights layout: layer groups at the root
    for got, want in zip(m2.get_weights(), _expected_list()):
        assert np.array_equal(got, want)
    with pytest.raises(ValueError, match="model_config"):
        networks.VxmDense.load(os.path.join(G, "keras_like_weights.h5"), device="cpu")
    other = networks.VxmDense((16, 16, 16), nb_unet_features=([4], [4, 4]), compute_dtype="fp32", device="cpu")
    with pytest.raises(ValueError, match="layers"):
        other.load_weights(os.path.join(G, "keras_like_vxm.h5"))
    other2 = networks.VxmDense((16, 16, 16), nb_unet_features=([4, 8], [6, 4, 4]), compute_dtype="fp32", device="cpu")
    with pytest.raises(ValueError, match="do not match"):
        other2.load_weights(os.path.join(G, "keras_like_vxm.h5"))


def test_vxmdense_save_h5_roundtrip(tmp_path):
    m = _tiny(seed=5)
    p = str(tmp_path / "0001.h5")
    m.save(p)
    m2 = networks.VxmDense.load(p, compute_dtype="fp32", device="cpu")
    assert m2.get_config() == m.get_config()
    for a, b in zip(m.get_weights(), m2.get_weights()):
        assert np.array_equal(a, b)
    pw = str(tmp_path / "w.h5")
    m.save_weights(pw)
    m3 = _tiny(seed=9)
    m3.load_weights(pw)
    for a, b in zip(m.get_weights(), m3.get_weights()):
        assert np.array_equal(a, b)
    m4 = _tiny(seed=9)
    before = m4.get_weights()
    m4.load_weights(pw, by_name=True)
    assert all(np.array_equal(a, b) for a, b in zip(m.get_weights(), m4.get_weights())) and not np.array_equal(before[0], m.get_weights()[0])
    # safetensors path is untouched
    ps = str(tmp_path / "m.safetensors")
    m.save(ps)
    m5 = networks.VxmDense.load(ps, compute_dtype="fp32", device="cpu")
    assert all(np.array_equal(a, b) for a, b in zip(m.get_weights(), m5.get_weights()))


@pytest.mark.skipif(not _have_h5py(), reason="no interpreter with h5py in this image")
def test_written_files_open_in_libhdf5(tmp_path):
    """What h5lite writes must be a valid HDF5 file for the real library, in the layout Keras' loader walks."""
    m = _tiny(seed=5)
    p = str(tmp_path / "0001.h5")
    m.save(p)
    np.savez(str(tmp_path / "w.npz"), *m.get_weights())
    root = h5lite.WGroup()  # a wide group (> 8 members -> several symbol-table nodes) and assorted dtypes
    for i in range(70):
        root.create_dataset(f"wide/d{i:03d}", np.full((3,), i, np.int32))
    root.create_dataset("types/f64", np.linspace(0, 1, 7))
    root.create_dataset("types/f16", np.linspace(0, 1, 7).astype(np.float16))
    root.create_dataset("types/u8", np.arange(9, dtype=np.uint8).reshape(3, 3))
    root.create_dataset("types/scalar", np.float32(1.25))
    root.create_dataset("types/empty", np.zeros((0, 2), np.float32))
    root.create_dataset("types/strs", np.array([b"abc", b"de"]))
    root.attrs["s"] = "text"
    root.attrs["vec"] = np.arange(3, dtype=np.int64)
    p2 = str(tmp_path / "misc.h5")
    h5lite.write_file(p2, root)
    code = f"""
import h5py, json, numpy as np
w = np.load({str(tmp_path / 'w.npz')!r})
with h5py.File({p!r}, 'r') as f:
    cfg = json.loads(f.attrs['model_config'].decode('utf8'))
    assert cfg['class_name'] == 'VxmDense' and cfg['config']['inshape'] == [16, 16, 16], cfg
    assert f.attrs['keras_version'] == b'2.7.0' and f.attrs['backend'] == b'tensorflow'
    g = f['model_weights']
    names = [n.decode('utf8') for n in g.attrs['layer_names']]
    assert len(names) == 6 and names[-1] == 'vxm_dense_flow' and names[0] == 'unet_enc_conv_0_0', names
    k = 0
    for n in names:
        wn = [x.decode('utf8') for x in g[n].attrs['weight_names']]
        assert wn == [n + '/kernel:0', n + '/bias:0'], wn
        for x in wn:
            assert np.array_equal(np.asarray(g[n][x]), w['arr_%d' % k]); k += 1
with h5py.File({p2!r}, 'r') as f:
    assert len(f['wide']) == 70 and all(f['wide/d%03d' % i][()].tolist() == [i] * 3 for i in range(70))
    assert f['types/f64'].dtype == np.float64 and np.allclose(f['types/f64'][()], np.linspace(0, 1, 7))
    assert f['types/f16'].dtype == np.float16 and f['types/u8'][()].tolist() == [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
    assert f['types/scalar'][()] == 1.25 and f['types/empty'].shape == (0, 2)
    assert f['types/strs'][()].tolist() == [b'abc', b'de']
    assert f.attrs['s'] == b'text' and f.attrs['vec'].tolist() == [0, 1, 2]
print('ok')
"""
    r = subprocess.run([H5PY_PYTHON, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr
