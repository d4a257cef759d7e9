"""Idle-gap analysis of a rocprofv3 --kernel-trace CSV: GPU busy fraction over the window holding the last N launches of the
dominant GEMM, and the gaps attributed to the kernel that FOLLOWS them.   python tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# timed region = from the first to the last attn_kernel<64 (ViT) launch in the second half of the trace
idx = [i for i, r in enumerate(rows) if "attn_kernel<64" in r["Kernel_Name"]]
idx = idx[len(idx) // 2:]
lo, hi = idx[0], idx[-1]
sel = rows[lo:hi + 1]
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
print(f"{len(sel)} launches, span {span/1e6:.3f} ms, kernel time {busy/1e6:.3f} ms, idle {100*(span-busy)/span:.1f}%")
gaps = collections.defaultdict(lambda: [0, 0])
for a, b in zip(sel[:-1], sel[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    k = b["Kernel_Name"][:60]
    gaps[k][0] += g; gaps[k][1] += 1
for k, (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  gap before {k:60s} n={n:5d} avg {t/n/1e3:7.2f} us total {t/1e6:7.3f} ms")
dur = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = r["Kernel_Name"][:60]; dur[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); dur[k][1] += 1
for k, (t, n) in sorted(dur.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  time in    {k:60s} n={n:5d} avg {t/n/1e3:7.2f} us total {t/1e6:7.3f} ms")
