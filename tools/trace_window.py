"""One window of a rocprofv3 --kernel-trace CSV, launch by launch: from the second-to-last to the last launch of a marker kernel.
    python tools/trace_window.py <trace dir> <marker kernel name substring> [marker launches per window, default 1]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
a, b = idx[-1 - per], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a + 1:b + 1] if sys.argv[2] != "preprocess_kernel" else rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  grid {r.get('Grid_Size_X', '?'):>7}x{r.get('Grid_Size_Y', '?'):>3} wg {r.get('Workgroup_Size_X', '?'):>4}  {r['Kernel_Name'][:100]}")
print(f"{len(rows[a:b])} launches, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
