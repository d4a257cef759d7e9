cd /tmp && export TMPDIR=/tmp
for F in 7 8 14 16; do
  rm -rf /tmp/tk$F; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk$F -- python /root/repo/tools/tick_bench.py $F 60 > /tmp/tk$F.log 2>&1
  echo "== frames $F"; python - <<PY
import csv, glob
f = glob.glob("/tmp/tk$F/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:9]:
    print(f'{r["Name"][:78]:78s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:8.1f} pct {float(r["Percentage"]):5.1f}')
PY
done
