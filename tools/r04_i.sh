#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm256.py -x -q -m gpu > $O/ops.log 2>&1; tail -5 $O/ops.log
for F in 1 4 8; do timeout 300 python tools/tick_bench.py $F 200 2>&1 | tail -1; done | tee $O/tick_latency.txt
for sm in 4 8; do echo "SMAX=$sm"; SM_POST_LN_SMAX=$sm timeout 300 python tools/tick_bench.py 1 200 2>&1 | tail -1; done
SM_POST_LN_FUSE=0 timeout 300 python tools/tick_bench.py 1 200 2>&1 | tail -1
timeout 1200 python -m pytest tests/test_gpu_path.py -x -q -m gpu > $O/path.log 2>&1; tail -5 $O/path.log
