#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04r; mkdir -p $O
echo "== conn_gate golden + mamba ops"; timeout 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_ops.py -x -q -m gpu -k "conn_gate_full_size_golden or mamba" 2>&1 | tail -2
echo "== planted"; timeout 1500 python -m pytest tests/test_gpu_parity_full.py -x -q -s -m gpu -k "planted" > $O/planted.log 2>&1; grep -E "flip at|planted|random|passed|failed" $O/planted.log | cut -c1-400
echo "== recurrent 1800"; timeout 1500 python -m pytest tests/test_gpu_parity_full.py -x -q -s -m gpu -k "3600" > $O/rec.log 2>&1; grep -E "^T=|passed|failed|assert" $O/rec.log | cut -c1-500
