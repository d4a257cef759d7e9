#!/bin/bash
# Build the library of a git revision (default HEAD) next to the working-tree one, for same-box A/B runs:
#   tools/build_base.sh [rev]  ->  streammind_amd/lib/libstreammind_hip_base.so   (select with STREAMMIND_HIP_LIB=...)
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C "$ROOT" archive "$REV" streammind_amd/csrc include | tar -x -C "$T"
mkdir -p "$T/o"
cd "$T/streammind_amd/csrc"
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -c "$f" -o "$T/o/${f%.hip}.o" & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/streammind_amd/lib/libstreammind_hip_base.so" "$T"/o/*.o
rm -rf "$T"
echo "$ROOT/streammind_amd/lib/libstreammind_hip_base.so"
