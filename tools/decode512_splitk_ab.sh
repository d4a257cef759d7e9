#!/bin/bash
# Same-box A/B: q|k|v / o_proj / down_proj of a 257..512-stream decode step as split-K slabs on the 256 x 256 kernel (thresholds by environment) against the
# 128 x 128 kernel they run on by default.  tools/group_decode_bench.py prints tokens/s and ms per step per group size.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
cd $R; mkdir -p gpurun_out/r06dec
SIZES=${SIZES:-384,512}
for rep in 1 2; do
  echo -n "default                    "; python tools/group_decode_bench.py $SIZES 2>&1 | tail -1 | cut -c1-400
  for cfg in "257 32 8" "257 32 4" "257 32 6"; do
    set -- $cfg
    echo -n "minrows=$1 mintiles=$2 smax=$3 "; SM_GEMM256_SPLITK_MINROWS=$1 SM_GEMM256_SPLITK_MINTILES=$2 SM_GEMM256_SPLITK_SMAX=$3 python tools/group_decode_bench.py $SIZES 2>&1 | tail -1 | cut -c1-400
  done
done | tee gpurun_out/r06dec/decode512_splitk_ab.txt
