#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python tools/pass_bench.py 28 40 > /dev/null 2>&1; grep -h "mamba" $(find /tmp/pp -name '*kernel_stats.csv' | head -1) | sed 's/.*SmSegStates, float\*)"//' | cut -c1-100
