#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round; mkdir -p $O
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
