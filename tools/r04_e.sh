#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04e; mkdir -p $O
for r in 0 1; do
rm -rf /tmp/prof_t$r; SM_GEMM_RING=$r timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t$r -- python tools/tick_bench.py 1 30 > $O/tick_ring$r.log 2>&1
f=$(find /tmp/prof_t$r -name '*kernel_trace.csv' | head -1)
python - "$f" > $O/layer_ring$r.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last tick: find the last preprocess_kernel
idx = [i for i, r in enumerate(rows) if "preprocess_kernel" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = t0
for r in rows[idx:idx + 330]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}  grid {r.get('Grid_Size_X','?'):>7} wg {r.get('Workgroup_Size_X','?'):>4}  {r['Kernel_Name'][:70]}")
    prev_end = e
PY
done
sed -n 1,45p $O/layer_ring1.txt
