#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04z; mkdir -p $O
rm -rf /tmp/pg; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python tools/group_decode_bench.py 128 > $O/g128.log 2>&1
cp "$(find /tmp/pg -name '*kernel_stats.csv' | head -1)" $O/group128_kernel_stats.csv; head -16 $O/group128_kernel_stats.csv | cut -c1-170
