#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_jpeg.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-300
