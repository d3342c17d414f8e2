#!/bin/bash
# Build a variant of the CURRENT library with extra -D flags: tools/ab/build_variant.sh <name> <flags...> -> tools/ab/libmmr_hip_<name>.so
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
tmp=$(mktemp -d)
cd "$root/multimodal-registration_amd/csrc"
objs=""
for f in api tail losses conv3d train synth eval hostio; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC "$@" -c $f.hip -o $tmp/$f.o 2>/dev/null &
  objs="$objs $tmp/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$root/tools/ab/libmmr_hip_$name.so"
rm -rf "$tmp"
echo built "$root/tools/ab/libmmr_hip_$name.so"
