#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm256.py -x -q -m gpu > $O/ops.log 2>&1; tail -3 $O/ops.log
for c in "COLD=0" "COLD=1 WARMX=1"; do for r in 0 1; do echo $c RING=$r; env $c STREAMMIND_HIP_LIB=$PWD/streammind_amd/lib/libstreammind_hip_tl.so SM_GEMM_RING=$r SM_SPLITK=1 python tools/gemm128_timeline.py 577 2>&1 | grep -v amdgpu.ids | cut -c1-330; done; done
STREAMMIND_HIP_LIB=$PWD/streammind_amd/lib/libstreammind_hip_tl.so SM_GEMM_RING=1 SM_SPLITK=1 python tools/gemm128_timeline.py 2308 2>&1 | grep -v amdgpu.ids | cut -c1-330
for r in 0 1; do for F in 1 4 8; do SM_GEMM_RING=$r timeout 300 python tools/tick_bench.py $F 200 2>&1 | tail -1; done; done
