#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_path.py -x -q -m gpu -k "mamba or conn_gate or stream_tiny or golden" 2>&1 | tail -3 | cut -c1-300
for R in 28 1; do timeout 300 python tools/pass_bench.py $R 200 2>&1 | tail -1; done
rm -rf /tmp/pp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python tools/pass_bench.py 28 40 > /dev/null 2>&1; grep -h "mamba" $(find /tmp/pp -name '*kernel_stats.csv' | head -1) | cut -c1-160
