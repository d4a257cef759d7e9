#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm256.py -x -q -m gpu > $O/ops.log 2>&1; tail -3 $O/ops.log
for r in 0 1; do echo "== SM_GEMM_RING=$r"; SM_GEMM_RING=$r SM_SPLITK=1 timeout 300 python tools/gemm_small_sweep.py worker 2>&1 | tr '\n' ' '; echo; SM_GEMM_RING=$r timeout 300 python tools/gemm_small_sweep.py worker 2>&1 | tr '\n' ' '; echo; done
for r in 0 1; do for F in 1 4 8; do SM_GEMM_RING=$r timeout 300 python tools/tick_bench.py $F 200 2>&1 | tail -1; done; done
