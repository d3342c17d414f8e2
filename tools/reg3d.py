#!/usr/bin/env python
"""Command-line twin of the reference's 3d_reg.py (same flags) on the MI355X engine.

python tools/reg3d.py --model-path m.h5 --config-path config_inference.json \
       --fx-img-path fixed.nii.gz --mov-img-path moving.nii.gz [--res-dir res] [--warp-interp linear]
       [--resample-interp linear] [--out-img-name warped_im] [--def-field-name deform_field]
Extra: --model-path-2 (cascade of bids_two_steps_registration.py), --compute-dtype bf16|fp32|fp32x3.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--model-path", required=True, type=str, help="path to the registration model (Keras .h5 or .safetensors)")
    p.add_argument("--model-path-2", default=None, help="optional second model (two-step cascade)")
    p.add_argument("--config-path", required=True, type=str, help="inference config (config_inference.json schema)")
    p.add_argument("--fx-img-path", required=True, help="path to the fixed image")
    p.add_argument("--mov-img-path", required=True, help="path to the moving image")
    p.add_argument("--res-dir", default="res", help="results output directory (default: res)")
    p.add_argument("--warp-interp", default="linear", help="linear or nearest (default: linear)")
    p.add_argument("--resample-interp", default="linear", help="linear, spline or nearest (default: linear)")
    p.add_argument("--out-img-name", default="warped_im")
    p.add_argument("--def-field-name", default="deform_field")
    p.add_argument("--compute-dtype", default="fp32x3", choices=["bf16", "fp32", "fp32x3"])
    a = p.parse_args()
    with open(a.config_path) as f:
        specs = json.load(f)
    from mmr import registration
    registration.run_3d_reg(specs, a.model_path, a.fx_img_path, a.mov_img_path, a.res_dir, a.warp_interp,
                            a.resample_interp, a.out_img_name, a.def_field_name, a.compute_dtype, a.model_path_2)


if __name__ == "__main__":
    main()
