#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p; mkdir -p $O
{
echo "baseline"; timeout 300 python tools/tick_bench.py 1 200 2>&1 | tail -1
echo "HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/tick_bench.py 1 200 2>&1 | tail -1
echo "HIP_FORCE_DEV_KERNARG=0"; HIP_FORCE_DEV_KERNARG=0 timeout 300 python tools/tick_bench.py 1 200 2>&1 | tail -1
echo "GPU_MAX_HW_QUEUES=1"; GPU_MAX_HW_QUEUES=1 timeout 300 python tools/tick_bench.py 1 200 2>&1 | tail -1
echo "HSA_ENABLE_SDMA=0"; HSA_ENABLE_SDMA=0 timeout 300 python tools/tick_bench.py 1 200 2>&1 | tail -1
echo "decode baseline"; timeout 300 python tools/decode_bench.py 64 1024 2>&1 | tail -1
echo "decode HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/decode_bench.py 64 1024 2>&1 | tail -1
} 2>&1 | tee $O/env_knobs.txt
timeout 120 tools/bin/grid_barrier_ubench 2>&1 | tee $O/grid_barrier.txt
