#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/ops.log 2>&1; tail -4 $O/ops.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "sliding or llm_tiny or group_batched or full_width_llm" > $O/path_sel.log 2>&1; tail -4 $O/path_sel.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_loader.py tests/test_gpu_graph.py -x -q -m gpu > $O/loader.log 2>&1; tail -4 $O/loader.log | cut -c1-300
for R in 28 1; do timeout 300 python tools/pass_bench.py $R 200 2>&1 | tail -1; done | tee $O/pass_bench.txt
SM_GATE_TAIL=0 timeout 300 python tools/pass_bench.py 28 200 2>&1 | tail -1
for F in 1 4 8; do timeout 300 python tools/tick_bench.py $F 200 2>&1 | tail -1; done | tee $O/tick_latency.txt
