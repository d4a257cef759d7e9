#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_jpeg.py tests/test_gpu_comm.py -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-aux --no-e2e --no-fp8 2>/dev/null | cut -c1-300
