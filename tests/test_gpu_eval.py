"""Device evaluation metrics vs golden vectors produced by running the reference's eval_reg_*.py scripts."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_metrics.npz"))


def test_jacobian_golden(dev):
    from mmr import evaluation as E
    for i in range(int(G["n_jac"])):
        r = E.jacobian_determinant(G[f"jac{i}_ddf"])
        np.testing.assert_allclose(r["det"], G[f"jac{i}_det"], rtol=1e-12, atol=1e-13)
        perc, med, mean, std, ntot, nneg = G[f"jac{i}_stats"]
        assert r["n_total"] == int(ntot) and r["n_negative"] == int(nneg)
        np.testing.assert_allclose([r["percentage_negative"], r["median"], r["mean"], r["std"]], [perc, med, mean, std],
                                   rtol=1e-12, atol=1e-13)
    ident = np.zeros((8, 8, 8, 1, 3))
    assert np.allclose(E.jacobian_determinant(ident)["det"], 1.0)


def test_nmi_golden(dev):
    from mmr import evaluation as E
    for i in range(3):
        got = E.normalized_mutual_information(G[f"nmi{i}_a"], G[f"nmi{i}_b"])
        np.testing.assert_allclose(got, float(G[f"nmi{i}_val"]), rtol=1e-12)
    assert E.detect_zero_padding(G["zp_im"]) == tuple(int(v) for v in G["zp_box"])
    got = E.nmi_report(G["nmi_script_fx"], G["nmi_script_mv"], G["nmi_script_wr"])
    np.testing.assert_allclose(got, G["nmi_script_vals"], rtol=1e-12)


def test_overlap_golden(dev):
    from mmr import evaluation as E
    hdr = [str(h) for h in G["seg_header"]]
    vals = dict(zip(hdr, G["seg_vals"]))
    before = E.overlap_metrics(G["seg_fx"], G["seg_mv"])
    after = E.overlap_metrics(G["seg_fx"], G["seg_wr"])
    for key, name in (("dice", "Dice"), ("jaccard", "Jaccard"), ("sensitivity", "Sensitivity"), ("precision", "Precision"),
                      ("specificity", "Specificity"), ("accuracy", "Accuracy")):
        b = [h for h in hdr if h.startswith(name) and "before" in h][0]
        a = [h for h in hdr if h.startswith(name) and "after" in h][0]
        np.testing.assert_allclose(before[key], vals[b], rtol=1e-12)
        np.testing.assert_allclose(after[key], vals[a], rtol=1e-12)
