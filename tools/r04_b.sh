#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04b; mkdir -p $O
timeout 120 tools/bin/stage_ubench > $O/stage_ubench.txt 2>&1; cat $O/stage_ubench.txt
timeout 2400 python -m pytest tests/test_gpu_parity_full.py -x -q -s -m gpu > $O/parity_full.log 2>&1; grep -E "^T=|steps, worst|56 frames|FLIPS|planted chain|passed|failed|Error|assert" $O/parity_full.log | cut -c1-700
