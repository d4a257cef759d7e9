#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04ac; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "group" 2>&1 | tail -3 | cut -c1-200
for f in 1 0; do echo "SM_POST_LN_FUSE=$f"; SM_POST_LN_FUSE=$f timeout 900 python tools/group_decode_bench.py 48,64,128 2>&1 | tail -1 | cut -c1-400; done | tee $O/rows_fuse.txt
