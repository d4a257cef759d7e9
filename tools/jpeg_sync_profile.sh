#!/bin/bash
# kernel stats of the self-synchronising JPEG decode (28 x 720p, Q = quality); summary -> gpurun_out/jsync_q$Q.txt
cd /tmp && export TMPDIR=/tmp
for Q in ${QS:-85}; do
  rm -rf /tmp/js$Q
  Q=$Q rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/js$Q -- python /root/repo/tools/jpeg_sync_trace.py > /tmp/js$Q.log 2>&1
  grep "ms per batch" /tmp/js$Q.log
  python - <<PY | tee /root/repo/gpurun_out/jsync_q$Q.txt
import csv, glob
f = glob.glob("/tmp/js$Q/**/*kernel_stats.csv", recursive=True)[0]
print(open("/tmp/js$Q.log").read().strip().splitlines()[-1])
for r in list(csv.DictReader(open(f)))[:14]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} pct {r["Percentage"]}')
PY
done
