#!/bin/bash
# Build round 4's library (git 09a1874) next to the current one for same-box A/B runs of single kernels (tools/ab_kernels.py).
# The result tools/ab/libmmr_hip_r04.so is git-ignored; it travels to the GPU box with the snapshot.
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive 09a1874 multimodal-registration_amd/csrc include | tar -x -C "$tmp"
cd "$tmp/multimodal-registration_amd/csrc"
objs=""
for f in api tail losses conv3d train synth eval; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -c $f.hip -o $f.o 2>/dev/null &
  objs="$objs $f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$root/tools/ab/libmmr_hip_r04.so"
rm -rf "$tmp"
echo built "$root/tools/ab/libmmr_hip_r04.so"
