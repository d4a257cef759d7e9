#!/bin/bash
# Per-call latency per frame count with the frame-lane rule (default), without it (SM_VIT_SMALL_LANES=1), and the equality test.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
cd $R; mkdir -p gpurun_out/r06cliff
timeout 900 python -m pytest tests/test_gpu_path.py -q -x -k "two_frame_lanes or two_tower_lanes or two_lanes_of_28 or three_lanes" 2>&1 | tail -2
for rep in 1 2; do for L in rule 1; do for F in 1 4 7 8 9 10 11 12 14 15 16 17 18 19 20 21 24 28; do
  echo -n "lanes=$L "; if [ $L = rule ]; then python tools/tick_bench.py $F 40 2>/dev/null | tail -1; else SM_VIT_SMALL_LANES=1 python tools/tick_bench.py $F 40 2>/dev/null | tail -1; fi
done; done; done | tee gpurun_out/r06cliff/small_lanes_rule_ab.txt
