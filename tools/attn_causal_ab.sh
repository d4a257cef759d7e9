#!/bin/bash
# Same-box A/B of the causal prefill attention (tools/attn_causal_bench.py): base library (tools/build_base.sh: the previous revision's tile kernel) next to the
# working tree's prefill kernel (SM_ATTN_PREFILL=1, default) and tile kernel under its three block schedules (SM_ATTN_PREFILL=0 SM_ATTN_PAIR=1|0|2); then the
# whole 2048-token prefill with either kernel.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
cd $R; mkdir -p gpurun_out/r06attn
{
for rep in 1 2; do
  for shape in "2048 0" "1024 0" "512 0" "2048 2048" "4096 0" "300 0"; do
    STREAMMIND_HIP_LIB=$R/streammind_amd/lib/libstreammind_hip_base.so SM_ATTN_PAIR=1 python tools/attn_causal_bench.py $shape 2>/dev/null | sed 's/^/base  /'
    for P in 1 2; do SM_ATTN_PREFILL=0 SM_ATTN_PAIR=$P python tools/attn_causal_bench.py $shape 2>/dev/null | sed 's/^/new   /'; done
    SM_CHECK=$([ $rep = 1 ] && echo 1 || echo 0) SM_ATTN_PREFILL=1 python tools/attn_causal_bench.py $shape 2>/dev/null | sed 's/^/new   /'
  done
done
} 2>&1 | tee gpurun_out/r06attn/attn_causal_ab2.txt
for rep in 1 2; do for P in 0 1; do echo -n "SM_ATTN_PREFILL=$P "; SM_ATTN_PREFILL=$P python tools/prefill_scan.py 512,2048 2>/dev/null | tail -2 | tr '\n' ' '; echo; done; done | tee gpurun_out/r06attn/prefill_ab2.txt
