#!/usr/bin/env python
"""Command-line twin of the reference's ``train_synthmorph.py --config-path config/config.json``.

One process per GPU:  python tools/train.py --config-path config.json
                      python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py --config-path config.json
The JSON is the reference's 44-key training config (config/config.json); the ``gpu`` key is ignored in favour of the
launcher's LOCAL_RANK.  Extra: --compute-dtype fp32x3|fp32, --checkpoint-ext .h5|.safetensors."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--config-path", default="config/config.json", help="config file with the training parameters specified")
    p.add_argument("--compute-dtype", default="fp32x3", choices=["fp32x3", "fp32"])
    p.add_argument("--checkpoint-ext", default=".h5", choices=[".h5", ".safetensors"])
    p.add_argument("--seed", type=int, default=0)
    a = p.parse_args()
    with open(a.config_path) as f:
        cfg = json.load(f)
    import torch
    from mmr import parallel, training
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rank, world, _ = parallel.init_from_env(device=dev)
    out = training.run_training(cfg, device=dev, rank=rank, world_size=world, seed=a.seed, compute_dtype=a.compute_dtype,
                                checkpoint_ext=a.checkpoint_ext)
    if out is not None and rank == 0:
        _, hist = out
        print(json.dumps({"epochs": len(hist), "last": hist[-1] if hist else None}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
