#!/bin/bash
# Per-kernel PMC averages for one command: tools/pmc_kernel.sh <kernel-regex> -- <cmd...>     (run on the GPU box)
# Each counter group is its own rocprofv3 run (--pmc only, no tracing), as the pool requires.
set -u
re="$1"; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA"
 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES"
 "SQ_VALU_MFMA_COEXEC_CYCLES SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU2 SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_CYCLES"
 "GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
)
n=0
for g in "${groups[@]}"; do
  rm -rf /tmp/pmc_$n
  rocprofv3 --pmc $g -d /tmp/pmc_$n --output-format csv -- "$@" > /tmp/pmc_$n.log 2>&1
  n=$((n+1))
done
python - "$re" <<'PY'
import csv, glob, re, sys, collections
rx = re.compile(sys.argv[1])
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if rx.search(r["Kernel_Name"]):
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{k:32s} {acc[k][0] / acc[k][1]:16.1f}  (n={acc[k][1]})")
PY
