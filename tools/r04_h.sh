#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04h; mkdir -p $O
SM_VIT_SMALL_LANES=1 timeout 300 python tools/tick_host_vs_gpu.py 1 2>&1 | grep -v amdgpu.ids | tee $O/host_vs_gpu.txt
timeout 300 python tools/lanes_threads_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/lanes_threads.txt
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/lanes_threads_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/lanes_threads_q8.txt
