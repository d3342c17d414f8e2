#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: every rocprofv3 pass the committed summaries under profiles/
# come from.  Usage: bash tools/profile_round.sh <tag>   (writes gpurun_out/<tag>_*).  Counters are collected in their
# own runs (one --pmc set per run, with --kernel-trace only), the program directly after `--` (no wrappers).
set -o pipefail
tag=${1:-r02}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step() { echo "[profile_round] $*"; }
# build OUTSIDE the profiler (hipcc chain-execs clang / lld, which must not happen under rocprofv3's preload), then pin the
# profiled runs to the built library so that none of them can start a build
python3 multimodal-registration_amd/build.py > $out/${tag}_build.log 2>&1 || exit 1
export MMR_LIB="$GRAFT_REPO_ROOT/multimodal-registration_amd/csrc/libmmr_hip.so"
# 1. kernel-trace stats of the exact default bench command (headline line with roofline, cpu_baseline, secondary)
step "stats: default bench"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_infer -- python3 bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err || exit 1
step "stats: train"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_train -- python3 bench.py --workload train --steps 10 --warmup 3 > $out/${tag}_bench_train.json 2> $out/${tag}_bench_train.err || exit 1
step "stats: ncc"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_ncc -- python3 bench.py --workload ncc --steps 20 --warmup 3 > $out/${tag}_bench_ncc.json 2> $out/${tag}_bench_ncc.err || exit 1
# 2. HBM traffic (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: separate passes)
for wl in infer ncc train; do
  extra="--no-cpu-baseline"; [ $wl = infer ] && extra="--no-cpu-baseline --no-secondary"
  for c in FETCH_SIZE WRITE_SIZE; do
    step "pmc $c: $wl"
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_${wl}_$c -- python3 bench.py --workload $wl --steps 3 --warmup 1 $extra > /dev/null 2> $out/${tag}_pmc_${wl}_$c.err || exit 1
  done
done
# 3. matrix-core / LDS / clock counters for the inference and training convs
for wl in infer train; do
  extra="--no-cpu-baseline"; [ $wl = infer ] && extra="--no-cpu-baseline --no-secondary"
  step "pmc SQ: $wl"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_pmc_${wl}_SQ -- python3 bench.py --workload $wl --steps 3 --warmup 1 $extra > /dev/null 2> $out/${tag}_pmc_${wl}_SQ.err || exit 1
done
# 4. the NCC / bending kernels are VALU / LDS machines, not matrix-core ones: instruction counts and VALU-active cycles
step "pmc SQ: ncc"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_pmc_ncc_SQ -- python3 bench.py --workload ncc --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/${tag}_pmc_ncc_SQ.err || exit 1
# 5. does the x2 FETCH_SIZE correction apply to the NCC kernel's loads (raw_buffer_load_b128)?  kernels of KNOWN traffic
if [ -x tools/ubench/fetch_calib ]; then
  step "pmc FETCH_SIZE: calibration kernels"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_calib -- ./tools/ubench/fetch_calib > $out/${tag}_fetch_calib.txt 2> $out/${tag}_fetch_calib.err || exit 1
fi
step done
