#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm256.py -x -q -m gpu > $O/ops.log 2>&1; tail -4 $O/ops.log | cut -c1-300
for F in 1 4 8; do timeout 300 python tools/tick_bench.py $F 200 2>&1 | tail -1; done | tee $O/tick_latency.txt
timeout 1500 python -m pytest tests/test_gpu_path.py -x -q -m gpu > $O/path.log 2>&1; tail -4 $O/path.log | cut -c1-300
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04k/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
print("per_call_latency", d.get("per_call_latency"))
print("conn_gate", d.get("rooflines_other", {}).get("connector_gate_pass"))
print("decode", {k: v for k, v in d.get("decode", {}).items() if not isinstance(v, (dict, list))})
PY
