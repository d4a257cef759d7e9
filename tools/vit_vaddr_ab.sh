cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r06pv
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "vit_attention or attention_random" 2>&1 | tail -2
{
for rep in 1 2 3; do for V in base new; do
  L=$R/streammind_amd/lib/libstreammind_hip.so; [ $V = base ] && L=$R/streammind_amd/lib/libstreammind_hip_base.so
  echo -n "$V "; STREAMMIND_HIP_LIB=$L python tools/attn_bench.py 28 2>/dev/null | tr '\n' ' '; echo
done; done
for rep in 1 2 3; do for V in base new; do
  L=$R/streammind_amd/lib/libstreammind_hip.so; [ $V = base ] && L=$R/streammind_amd/lib/libstreammind_hip_base.so
  for B in 28 56; do echo -n "$V batch $B "; STREAMMIND_HIP_LIB=$L python bench.py --batch $B --steps 24 --warmup 3 --no-cpu-baseline --no-decode --no-aux --no-e2e --no-fp8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print(d['value'], d['ms_per_step'])"; done
done; done
} 2>&1 | tee gpurun_out/r06pv/vit_vaddr_ab.txt
