set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py > gpurun_out/r01t_bench_c2_infer.json
python bench.py --workload train --steps 8 --warmup 2 > gpurun_out/r01t_bench_c3_train_fp32x3.json
python bench.py --workload train --dtype fp32 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r01t_bench_c3_train_fp32.json
python bench.py --workload train --bwd bf16 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r01t_bench_c3_train_bwd_bf16.json
python bench.py --workload ncc --steps 50 --warmup 5 > gpurun_out/r01t_bench_c5_ncc.json
python bench.py --workload cascade --steps 5 --warmup 1 > gpurun_out/r01t_bench_c4_cascade.json
python bench.py --dtype fp32x3 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r01t_bench_c2_infer_fp32x3.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01t_infer -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01t_train -- python3 $GRAFT_REPO_ROOT/bench.py --workload train --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
echo done
