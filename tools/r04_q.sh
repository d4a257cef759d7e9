#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04q; mkdir -p $O
echo "== conn_gate golden, default"; timeout 600 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "conn_gate_full_size_golden" 2>&1 | tail -2
echo "== SM_GATE_TAIL=0"; SM_GATE_TAIL=0 timeout 600 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "conn_gate_full_size_golden" 2>&1 | tail -2
echo "== planted"; timeout 1500 python -m pytest tests/test_gpu_parity_full.py -x -q -s -m gpu -k "planted" > $O/planted.log 2>&1; grep -E "planted|random|passed|failed|FLIPS|flip" $O/planted.log | cut -c1-400
bash tools/r04_p.sh
