#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "group" > $O/group.log 2>&1; tail -6 $O/group.log | cut -c1-400
timeout 900 python tools/group_decode_bench.py 32,48,64,96,128 2>&1 | tail -1 | tee $O/group_decode.txt
