#!/bin/bash
# round 4 (re-entry): one-frame tick kernel trace + per-launch timeline, default bench line
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04f; mkdir -p $O
for F in 1 4 8; do timeout 300 python tools/tick_bench.py $F 200 2>&1 | tail -1; done | tee $O/tick_latency.txt
rm -rf /tmp/prof_t1; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t1 -- python tools/tick_bench.py 1 200 > $O/tick_b1_profiled.log 2>&1
cp "$(find /tmp/prof_t1 -name '*kernel_stats.csv' | head -1)" $O/tick_b1_kernel_stats.csv
python tools/stats_per_step.py /tmp/prof_t1 210 > $O/tick_b1_per_step.txt 2>&1; cut -c1-200 $O/tick_b1_per_step.txt
f=$(find /tmp/prof_t1 -name '*kernel_trace.csv' | head -1)
python - "$f" > $O/tick_b1_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "preprocess_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}  grid {r.get('Grid_Size_X','?'):>7} wg {r.get('Workgroup_Size_X','?'):>4}  {r['Kernel_Name'][:90]}")
    prev_end = e
PY
wc -l $O/tick_b1_timeline.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-1500 $O/bench_default.json
