#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04ai; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_path.py tests/test_gpu_loader.py tests/test_gpu_serve.py -x -q -m gpu -k "llm or prefill or teacher or decode or generate or loader or checkpoint or serve or stream" > $O/llm.log 2>&1; tail -4 $O/llm.log | cut -c1-300
for f in 1 0; do echo -n "SM_POST_LN_FUSE=$f "; SM_POST_LN_FUSE=$f timeout 300 python tools/decode_bench.py 16 1024 2>&1 | tail -1 | cut -c1-160; done | tee $O/prefill_ab.txt
