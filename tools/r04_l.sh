#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "post_ln or mamba" > $O/ops.log 2>&1; tail -3 $O/ops.log | cut -c1-300
for f in 1 0; do for R in 28 1 8; do echo -n "FUSE=$f "; SM_POST_LN_FUSE=$f timeout 300 python tools/pass_bench.py $R 200 2>&1 | tail -1; done; done | tee $O/pass_bench.txt
timeout 1500 python -m pytest tests/test_gpu_path.py -x -q -m gpu > $O/path.log 2>&1; tail -4 $O/path.log | cut -c1-300
