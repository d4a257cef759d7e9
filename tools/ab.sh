#!/bin/bash
# same-box A/B of the default bench: base library (tools/build_base.sh) vs the working-tree build, alternating, N rounds
N=${1:-3}; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq $N); do for lib in base new; do
  L=$R/streammind_amd/lib/libstreammind_hip.so; [ $lib = base ] && L=$R/streammind_amd/lib/libstreammind_hip_base.so
  STREAMMIND_HIP_LIB=$L python $R/bench.py --no-cpu-baseline --no-fp8 --no-e2e --no-decode "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], d['roofline']['achieved'] if d.get('roofline') else None)"
done; done
