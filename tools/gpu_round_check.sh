#!/bin/bash
# One gpurun call that re-checks a round on the GPU box: the -m gpu suite, smoke(), every bench workload, the 2-rank rehearsal
# and the A/B tools the DESIGN numbers come from.  Usage (from the repo root, through gpurun): bash tools/gpu_round_check.sh <tag>
set -e -o pipefail
tag=${1:-r03}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py > gpurun_out/${tag}_bench_default.json
python bench.py --workload train --steps 20 --warmup 3 > gpurun_out/${tag}_bench_train.json
python bench.py --workload train --dtype fp32 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_train_fp32.json
python bench.py --workload ncc --steps 50 --warmup 5 > gpurun_out/${tag}_bench_ncc.json
python bench.py --workload cascade --steps 5 --warmup 1 > gpurun_out/${tag}_bench_cascade.json
MMR_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/${tag}_bench_2ranks_gloo.json
python tools/time_upfold.py > gpurun_out/${tag}_time_upfold.txt 2>&1
python tools/time_c2_layers.py > gpurun_out/${tag}_time_c2_layers.txt 2>&1
python tools/time_tail_layers.py > gpurun_out/${tag}_time_tail_layers.txt 2>&1
python tools/hbm_ceiling.py > gpurun_out/${tag}_hbm_ceiling.txt 2>&1
python tools/time_ncc_overlap.py > gpurun_out/${tag}_time_ncc_overlap.txt 2>&1
MMR_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 3 --warmup 1 > gpurun_out/${tag}_bench_4ranks_gloo.json
[ -x tools/ubench/mfma_ceiling ] && ./tools/ubench/mfma_ceiling > gpurun_out/${tag}_mfma_ceiling.txt 2>&1
echo done
