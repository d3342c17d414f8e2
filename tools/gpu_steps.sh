#!/bin/bash
# Run a list of GPU steps inside ONE gpurun call: each step under its own `timeout -k 10`, output to gpurun_out/<tag>_<name>.log.
# A failed assertion lets the next step run; a step that was KILLED (timeout / signal) ends the call -- no further GPU step is
# started behind a hung one.  Usage: bash tools/gpu_steps.sh <tag> <file with lines "name|seconds|command">
tag=$1; list=$2
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
python3 multimodal-registration_amd/build.py > gpurun_out/${tag}_build.log 2>&1 || { echo "build failed"; tail -20 gpurun_out/${tag}_build.log; exit 1; }
while IFS='|' read -r name secs cmd; do
  [ -z "$name" ] && continue
  echo "[gpu_steps] $name: $cmd"
  t0=$(date +%s)
  timeout -k 10 "$secs" bash -o pipefail -c "$cmd" > gpurun_out/${tag}_${name}.log 2>&1
  rc=$?
  echo "[gpu_steps] $name rc=$rc $(( $(date +%s) - t0 ))s"; tail -4 gpurun_out/${tag}_${name}.log
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "[gpu_steps] $name was killed: stopping"; exit $rc; fi
done < "$list"
echo "[gpu_steps] done"
