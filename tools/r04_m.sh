#!/bin/bash
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04m; mkdir -p $O
tl() { # $1 = trace dir, $2 = marker kernel, $3 = out
python - "$1" "$2" > "$3" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  grid {r.get('Grid_Size_X','?'):>7}x{r.get('Grid_Size_Y','?'):>3} wg {r.get('Workgroup_Size_X','?'):>4}  {r['Kernel_Name'][:100]}")
PY
}
rm -rf /tmp/p28; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p28 -- python tools/pass_bench.py 28 40 > $O/p28.log 2>&1
tl /tmp/p28 gate_decide_kernel $O/pass28_timeline.txt; cat $O/pass28_timeline.txt
for F in 4 8; do
rm -rf /tmp/pf$F; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf$F -- python tools/tick_bench.py $F 60 > $O/tick_f$F.log 2>&1
python tools/stats_per_step.py /tmp/pf$F 70 > $O/tick_f${F}_per_step.txt 2>&1; head -14 $O/tick_f${F}_per_step.txt | cut -c1-160
done
tl /tmp/pf8 preprocess_kernel $O/tick_f8_timeline.txt; sed -n 1,40p $O/tick_f8_timeline.txt
