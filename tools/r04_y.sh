#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04y; mkdir -p $O
for t in 128 320; do for F in 4 6 8; do echo -n "MAXTILES=$t "; SM_POST_LN_MAXTILES=$t timeout 300 python tools/tick_bench.py $F 200 2>&1 | tail -1; done; done | tee $O/maxtiles.txt
