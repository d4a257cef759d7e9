#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04o; mkdir -p $O
timeout 3400 python -m pytest tests -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1; tail -30 $O/pytest_gpu.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
