#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04w; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_path.py -x -q -m gpu > $O/path.log 2>&1; tail -5 $O/path.log | cut -c1-300
for v in 1 0; do echo "SM_VIT_FC2_MEAN=$v"; SM_VIT_FC2_MEAN=$v timeout 600 python bench.py --no-cpu-baseline --no-decode --no-aux --no-e2e --no-fp8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'])"; SM_VIT_FC2_MEAN=$v timeout 300 python tools/tick_bench.py 1 200 2>&1 | tail -1; done | tee $O/fc2mean_ab.txt
for v in 1 0 1 0; do echo -n "FC2_MEAN=$v single lane: "; SM_VIT_FC2_MEAN=$v timeout 600 python bench.py --batch 28 --no-pipeline --no-cpu-baseline --no-decode --no-aux --no-e2e --no-fp8 --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; done | tee -a $O/fc2mean_ab.txt
