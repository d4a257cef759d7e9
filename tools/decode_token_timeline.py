"""One decode token from a rocprofv3 --kernel-trace CSV of tools/decode_bench.py: per-kernel average duration, the gap before it,
and their sum over the launches between two consecutive argmax kernels (= one token).  python tools/decode_token_timeline.py <dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
am = [i for i, r in enumerate(rows) if "argmax" in r["Kernel_Name"]]
am = am[len(am) // 2:]                      # steady state: second half of the tokens
tok = len(am) - 1
dur = collections.defaultdict(lambda: [0, 0, 0])
for a, b in zip(am[:-1], am[1:]):
    for i in range(a + 1, b + 1):
        r, p = rows[i], rows[i - 1]
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
        d = dur[k]
        d[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[1] += 1
        d[2] += int(r["Start_Timestamp"]) - int(p["End_Timestamp"])
span = (int(rows[am[-1]]["End_Timestamp"]) - int(rows[am[0]]["End_Timestamp"])) / tok / 1e3
print(f"{tok} tokens, {span:.1f} us per token")
tk = tg = 0
for k, (t, n, g) in sorted(dur.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:50s} {n / tok:6.1f}/tok  avg {t / n / 1e3:6.2f} us  gap before {g / n / 1e3:5.2f} us   = {t / tok / 1e3:7.1f} + {g / tok / 1e3:6.1f} us/tok")
    tk += t / tok / 1e3; tg += g / tok / 1e3
print(f"kernels {tk:.1f} us + gaps {tg:.1f} us")
