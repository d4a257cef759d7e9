#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "sliding" 2>&1 | tail -6 | cut -c1-400
