#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_jpeg.py -x -q -m gpu > $O/jpeg.log 2>&1; tail -15 $O/jpeg.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_comm.py -x -q -m gpu > $O/comm.log 2>&1; tail -25 $O/comm.log | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k post_ln > $O/ops.log 2>&1; tail -5 $O/ops.log | cut -c1-300
