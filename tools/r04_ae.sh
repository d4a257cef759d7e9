#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python tools/lane_priority_probe.py 2>&1 | grep priority | tee gpurun_out/lane_priority.txt
