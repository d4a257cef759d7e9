#!/bin/bash
# same-box A/B of the ViT GEMM shapes (tools/gemm_bench.py, M = 16156) across library variants:
#   tools/ab_gemm.sh ROUNDS name1 name2 ...     (name "" = the in-tree library, otherwise libstreammind_hip_<name>.so)
N=${1:-2}; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq $N); do for v in "$@"; do
  L=$R/streammind_amd/lib/libstreammind_hip.so; [ "$v" != main ] && L=$R/streammind_amd/lib/libstreammind_hip_$v.so
  echo "== $v round $i"
  STREAMMIND_HIP_LIB=$L python $R/tools/gemm_bench.py 16156 2>&1 | grep -v "^$"
done; done
