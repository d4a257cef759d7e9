#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "post_norm or repeated_column or post_ln" > $O/ops.log 2>&1; tail -6 $O/ops.log | cut -c1-300
