"""Per-step kernel time table from a rocprofv3 --kernel-trace csv: python tools/stats_per_step.py <dir> <steps_total>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = float(sys.argv[2])
d = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    d[k][0] += 1
    d[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{v[1] / steps / 1e3:9.1f} us/step  {v[0] / steps:7.1f} calls/step  {v[1] / v[0] / 1e3:7.2f} us avg  {k}")
print(f"total {tot / steps / 1e3:.1f} us/step")
