#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04g; mkdir -p $O
timeout 300 python tools/small_lanes_check.py 2>&1 | grep -v amdgpu.ids | tr '\n' ';'; echo
for L in 1 2 4 8; do for F in 2 4 8; do echo -n "lanes<=$L: "; SM_VIT_SMALL_LANES=$L timeout 300 python tools/tick_bench.py $F 200 2>&1 | tail -1; done; done | tee $O/small_lanes.txt
