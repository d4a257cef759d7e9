#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04x; mkdir -p $O
run() { timeout 600 python bench.py --batch 28 --no-pipeline --no-cpu-baseline --no-decode --no-aux --no-e2e --no-fp8 --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
run2() { timeout 600 python bench.py --no-cpu-baseline --no-decode --no-aux --no-e2e --no-fp8 --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
{
echo -n "base single: "; run
echo -n "OUT=256128 single: "; SM_VIT_OUT_TILE=256128 run
echo -n "FC2=256128 single: "; SM_VIT_FC2_TILE=256128 run
echo -n "OUT=128 single: "; SM_VIT_OUT_TILE=128 run
echo -n "base single: "; run
echo -n "base two lanes: "; run2
echo -n "OUT=256128 two lanes: "; SM_VIT_OUT_TILE=256128 run2
echo -n "OUT+FC2=256128 two lanes: "; SM_VIT_OUT_TILE=256128 SM_VIT_FC2_TILE=256128 run2
echo -n "base two lanes: "; run2
} | tee $O/tile_ab.txt
