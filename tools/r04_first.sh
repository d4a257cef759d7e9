#!/bin/bash
# round 4, first GPU call: new full-size parity tests + kernel trace of the one-frame tick
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_full.py -x -q -s -m gpu > $O/parity_full.log 2>&1; tail -25 $O/parity_full.log | cut -c1-600
for F in 1 4 8; do
  timeout 300 python tools/tick_bench.py $F 200 2>&1 | tail -1
done
rm -rf /tmp/prof_t1; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t1 -- python tools/tick_bench.py 1 200 > $O/tick_b1_profiled.log 2>&1
cp "$(find /tmp/prof_t1 -name '*kernel_stats.csv' | head -1)" $O/tick_b1_kernel_stats.csv
python tools/stats_per_step.py /tmp/prof_t1 210 > $O/tick_b1_per_step.txt 2>&1; cat $O/tick_b1_per_step.txt | cut -c1-200
