#!/bin/bash
# Build the WORKING TREE's library under another name with extra compiler flags, for same-box A/B runs:
#   tools/build_variant.sh v2 -DG2_DIRECT_EPI=1   ->  streammind_amd/lib/libstreammind_hip_v2.so  (select with STREAMMIND_HIP_LIB=...)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cd "$ROOT/streammind_amd/csrc"
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result "$@" -c "$f" -o "$T/${f%.hip}.o" & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/streammind_amd/lib/libstreammind_hip_$NAME.so" "$T"/*.o
rm -rf "$T"
echo "$ROOT/streammind_amd/lib/libstreammind_hip_$NAME.so"
