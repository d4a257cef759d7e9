#!/bin/bash
# Same-box A/B of the 128 x 128 ring kernel's prefetch wave (SM_GEMM128_PFW = k-tiles ahead, 0 = off): the LLM products with HBM-cold weights in isolation, batched decode,
# short prefills, per-call latency of small frame counts.  Every command under its own timeout (a barrier mismatch would hang the kernel).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
cd $R; mkdir -p gpurun_out/r06pfw
echo "== operator tests with the prefetch wave on"; SM_GEMM128_PFW=8 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gemm256.py -q -x -k "linear or gemm or tiled" 2>&1 | tail -2
{
for D in ${DS:-0 6 8 12 16}; do
  echo "== SM_GEMM128_PFW=$D cold products M=512"; SM_GEMM128_PFW=$D COLD=1 timeout 120 python tools/gemm_bench_llm.py 512 2>/dev/null | grep -v lm_head
  echo "== SM_GEMM128_PFW=$D cold products M=256"; SM_GEMM128_PFW=$D COLD=1 timeout 120 python tools/gemm_bench_llm.py 256 2>/dev/null | grep -v lm_head
done
for rep in 1 2; do for D in ${DS2:-0 8 12}; do
  echo -n "SM_GEMM128_PFW=$D decode  "; SM_GEMM128_PFW=$D timeout 300 python tools/group_decode_bench.py 160,256,512 2>/dev/null | tail -1 | cut -c1-330
  echo -n "SM_GEMM128_PFW=$D prefill "; SM_GEMM128_PFW=$D timeout 300 python tools/prefill_scan.py 160,328,512 2>/dev/null | tr '\n' ';'; echo
  for F in 1 2 4; do echo -n "SM_GEMM128_PFW=$D tick "; SM_GEMM128_PFW=$D timeout 200 python tools/tick_bench.py $F 60 2>/dev/null | tail -1; done
done; done
} 2>&1 | tee gpurun_out/r06pfw/pfw_ab.txt
