#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04u; mkdir -p $O
for mb in 12 0 40; do echo -n "SM_SKINNY_DIRECT_MB=$mb "; SM_SKINNY_DIRECT_MB=$mb timeout 300 python tools/pass_bench.py 28 200 2>&1 | tail -1; done | tee $O/direct_ab.txt
for mb in 12 0; do echo -n "group decode DIRECT_MB=$mb "; SM_SKINNY_DIRECT_MB=$mb timeout 600 python tools/group_decode_bench.py 32 2>&1 | tail -1 | cut -c1-300; done | tee -a $O/direct_ab.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_path.py -x -q -m gpu -k "skinny or conn_gate or group or stream" > $O/t.log 2>&1; tail -3 $O/t.log | cut -c1-200
