#!/bin/bash
# Same-box A/B of the ViT attention's PV phase: main (two transpose reads, a wait, two MFMAs, ten times per tile) vs library variants built with
# tools/build_variant.sh pv2 -DVIT_PV_BATCH=2 (eight reads of a 32-key block as a batch, 128 registers) and pv1 -DVIT_PV_BATCH=1 (all sixteen, 142 registers = 3 waves per SIMD)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
cd $R; mkdir -p gpurun_out/r06pv
{
for rep in 1 2 3; do for V in main pv2 pv1; do
  L=$R/streammind_amd/lib/libstreammind_hip.so; [ $V != main ] && L=$R/streammind_amd/lib/libstreammind_hip_$V.so
  echo -n "$V "; SM_CHECK=$([ $rep = 1 ] && echo 1 || echo 0) STREAMMIND_HIP_LIB=$L python tools/attn_bench.py 28 2>/dev/null | tr '\n' ' '; echo
done; done
for rep in 1 2; do for V in main pv2; do
  L=$R/streammind_amd/lib/libstreammind_hip.so; [ $V != main ] && L=$R/streammind_amd/lib/libstreammind_hip_$V.so
  for B in 28 56; do echo -n "$V batch $B "; STREAMMIND_HIP_LIB=$L python bench.py --batch $B --steps 24 --warmup 3 --no-cpu-baseline --no-decode --no-aux --no-e2e --no-fp8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print(d['value'], d['ms_per_step'])"; done
done; done
} 2>&1 | tee gpurun_out/r06pv/vit_pv_ab.txt
