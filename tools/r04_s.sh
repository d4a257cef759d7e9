#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "group" 2>&1 | tail -3
for f in 1 0; do echo "SM_POST_LN_FUSE=$f"; SM_POST_LN_FUSE=$f timeout 600 python tools/group_decode_bench.py 16,24,32 2>&1 | tail -1 | cut -c1-900; done | tee $O/group_decode.txt
rm -rf /tmp/pg; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python tools/group_decode_bench.py 32 > $O/g32.log 2>&1
cp "$(find /tmp/pg -name '*kernel_stats.csv' | head -1)" $O/group32_kernel_stats.csv; head -14 $O/group32_kernel_stats.csv | cut -c1-180
