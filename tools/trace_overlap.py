"""Concurrency in a rocprofv3 --kernel-trace CSV: for every kernel class, its launches' total duration and the part of it during
which a GEMM of ANOTHER stream (queue) was running -- i.e. whether the light row passes run underneath the other tower lane's
matrix kernels.  Also the union busy time and the time with >= 1 GEMM in flight over the second half of the trace.
    python tools/trace_overlap.py <kernel_trace.csv>"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
rows = rows[len(rows) // 2:]
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)


def cls(name):
    for k in ("gemm256p_kernel", "gemm256_kernel", "vit_attn_kernel", "ln_light_kernel", "norm_wave_fixed_kernel", "skinny", "pool16", "gemm_kernel"):
        if k in name:
            return k
    return "other"


gemms = [r for r in rows if cls(r["Kernel_Name"]).startswith("gemm256")]


def overlap_with_gemm(r):
    t = 0
    for g in gemms:
        if g["s"] >= r["e"]:
            break
        if g is r or (qkey and g[qkey] == r[qkey]):
            continue
        lo, hi = max(g["s"], r["s"]), min(g["e"], r["e"])
        if hi > lo:
            t += hi - lo
    return t


tot = collections.defaultdict(lambda: [0, 0, 0])
for r in rows:
    c = cls(r["Kernel_Name"])
    tot[c][0] += r["e"] - r["s"]; tot[c][1] += 1; tot[c][2] += overlap_with_gemm(r)
span = rows[-1]["e"] - rows[0]["s"]
ev = sorted([(r["s"], 1) for r in rows] + [(r["e"], -1) for r in rows])
busy = 0; n = 0; last = ev[0][0]
for t, d in ev:
    if n > 0:
        busy += t - last
    n += d; last = t
print(f"span {span/1e6:.3f} ms, >= 1 kernel in flight {100*busy/span:.1f} %, queues: {len(set(r[qkey] for r in rows)) if qkey else '?'}")
for c, (t, k, o) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"  {c:26s} n={k:5d} avg {t/k/1e3:8.2f} us  sum {t/1e6:8.3f} ms  of which beside another queue's gemm256: {100*o/max(t,1):5.1f} %")
