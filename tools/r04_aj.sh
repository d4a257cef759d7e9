#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04aj; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "skinny or post" > $O/ops.log 2>&1; tail -4 $O/ops.log | cut -c1-300
for v in 1 0; do echo -n "SM_GATE_NORM_IN=$v "; SM_GATE_NORM_IN=$v timeout 300 python tools/pass_bench.py 1 300 2>&1 | tail -1; done | tee $O/norm_in.txt
for v in 1 0; do echo -n "SM_GATE_NORM_IN=$v "; SM_GATE_NORM_IN=$v timeout 300 python tools/tick_bench.py 1 200 2>&1 | tail -1; done | tee -a $O/norm_in.txt
timeout 1500 python -m pytest tests/test_gpu_path.py tests/test_gpu_graph.py -x -q -m gpu > $O/path.log 2>&1; tail -4 $O/path.log | cut -c1-300
