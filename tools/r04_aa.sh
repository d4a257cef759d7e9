#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_serve.py -x -q -m gpu -k "lane or session or serve or worker" 2>&1 | tail -4 | cut -c1-300
