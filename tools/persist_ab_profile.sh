#!/bin/bash
# same-box A/B of the persistent 256x256 GEMM in the single-lane plain schedule: rocprofv3 kernel stats with SM_GEMM_PERSIST=0 / 1
# (-> gpurun_out/round/kernel_stats_steps_persist{0,1}.csv) and the PMC passes (traffic, MFMA busy) for both
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round; mkdir -p $O
B="python bench.py --batch 28 --no-pipeline --steps 20 --warmup 2 --no-cpu-baseline --no-decode --no-aux --no-e2e --no-prof"
for P in 0 1; do
  rm -rf /tmp/ps$P; SM_GEMM_PERSIST=$P rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps$P -- $B > $O/ab_persist$P.log 2>&1
  cp "$(find /tmp/ps$P -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_steps_persist$P.csv
  grep '^{"metric"' $O/ab_persist$P.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist=$P', d['value'], d['ms_per_step'])"
  head -5 $O/kernel_stats_steps_persist$P.csv | cut -c1-120
done
S="python bench.py --batch 28 --no-pipeline --steps 3 --warmup 1 --stream-frames 112 --no-cpu-baseline --no-decode --no-prof --no-aux --no-e2e"
for P in 0 1; do
  rm -rf /tmp/pf$P /tmp/pw$P /tmp/pm$P
  SM_GEMM_PERSIST=$P timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf$P -- $S > $O/pmc_f$P.log 2>&1; echo "persist=$P FETCH rc=$?"
  SM_GEMM_PERSIST=$P timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw$P -- $S > $O/pmc_w$P.log 2>&1; echo "persist=$P WRITE rc=$?"
  python tools/pmc_traffic_summary.py /tmp/pf$P /tmp/pw$P $O/gemm_traffic_persist$P.json 2>&1 | tail -2
  SM_GEMM_PERSIST=$P timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d /tmp/pm$P -- $S > $O/pmc_m$P.log 2>&1; echo "persist=$P MFMA rc=$?"
  python tools/pmc_mfma_summary.py /tmp/pm$P $O/mfma_util_persist$P.json 2>&1 | tail -2
done
tail -5 $O/pmc_f1.log
