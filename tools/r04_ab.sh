#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04ab; mkdir -p $O
for m in 8 4; do echo "SM_SPLITK_SMALLM_MAX=$m"; SM_SPLITK_SMALLM_MAX=$m timeout 900 python tools/group_decode_bench.py 48,64,128 2>&1 | tail -1 | cut -c1-400; done | tee $O/splitk_smallm.txt
for m in 8 4; do echo -n "SMALLM_MAX=$m prefill/decode: "; SM_SPLITK_SMALLM_MAX=$m timeout 300 python tools/decode_bench.py 32 1024 2>&1 | tail -1 | cut -c1-200; done | tee -a $O/splitk_smallm.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_path.py -x -q -m gpu -k "tiled_gemm or group or llm" 2>&1 | tail -3 | cut -c1-200
