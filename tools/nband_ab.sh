#!/bin/bash
# Same-box A/B of the 128 x 128 kernel's tile order (SM_GEMM128_NBAND=0: an XCD's band of tiles walks n fastest, 1: m fastest, unset: by shape): batched decode at
# 129..512 streams, short prefills, per-call latency of small frame counts.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
cd $R; mkdir -p gpurun_out/r06nband
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm256.py -q -x -k "linear or gemm" 2>&1 | tail -2
{
for rep in 1 2; do for NB in 0 1; do
  echo -n "SM_GEMM128_NBAND=$NB decode  "; SM_GEMM128_NBAND=$NB python tools/group_decode_bench.py 160,256,384,512 2>/dev/null | tail -1 | cut -c1-330
  echo -n "SM_GEMM128_NBAND=$NB prefill "; SM_GEMM128_NBAND=$NB python tools/prefill_scan.py 160,328,512,768 2>/dev/null | tr '\n' ';'; echo
  for F in 1 2 4 8 12; do echo -n "SM_GEMM128_NBAND=$NB tick "; SM_GEMM128_NBAND=$NB python tools/tick_bench.py $F 60 2>/dev/null | tail -1; done
done; done
} 2>&1 | tee gpurun_out/r06nband/nband_ab.txt
