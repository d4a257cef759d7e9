#!/bin/bash
# The LLM products at 129..512 rows with HBM-cold weights (COLD=1), per kernel configuration: which tile / ring / split-K choice streams them best.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
cd $R; mkdir -p gpurun_out/r06nband
for M in ${MS:-512 256}; do
  for cfg in "" "SM_GEMM128_NBAND=0" "SM_GEMM_RING=0" "SM_SPLITK=2" "SM_SPLITK=4" "SM_GEMM_TILE=256128" "SM_GEMM_TILE=256" "SM_GEMM_RING=2 SM_SPLITK=2"; do
    echo "== M=$M $cfg"; env COLD=1 $cfg python tools/gemm_bench_llm.py $M 2>/dev/null | grep -v lm_head
  done
done 2>&1 | tee gpurun_out/r06nband/gemm128_cold_ab.txt
