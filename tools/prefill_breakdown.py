"""Per-kernel time of the LAST prefill in a rocprofv3 --kernel-trace CSV of tools/decode_bench.py (the launches between the
last embed_splice kernel and the argmax that follows it).   python tools/prefill_breakdown.py <dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
es = [i for i, r in enumerate(rows) if "embed_splice" in r["Kernel_Name"]]
lo = es[-1]
hi = next(i for i in range(lo, len(rows)) if "argmax" in rows[i]["Kernel_Name"])
sel = rows[lo:hi + 1]
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
d = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:56]
    d[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[k][1] += 1
print(f"prefill: {len(sel)} launches, {span:.0f} us")
for k, (t, n) in sorted(d.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {k:58s} n={n:4d} avg {t / n / 1e3:7.2f} us  total {t / 1e3:8.1f} us")
