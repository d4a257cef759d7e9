#!/bin/bash
# same-box A/B of cache-policy variants of the streamed GEMM outputs (common.h store16_stream; tools/build_variant.sh NAME '-DSM_STREAM_STORE_MODS="sc1 nt"'):
#   gpurun -- 'VARIANTS="main nt1" BATCH=56 bash tools/nt_store_ab.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/${TAG:-r06k}
BENCH_FAST="--no-cpu-baseline --no-decode --no-aux --no-e2e --no-fp8"
for i in 1 2; do for v in ${VARIANTS:-main nt1 nt2 nt3}; do for B in ${BATCH:-28}; do
  L=$GRAFT_REPO_ROOT/streammind_amd/lib/libstreammind_hip.so; [ "$v" != main ] && L=$GRAFT_REPO_ROOT/streammind_amd/lib/libstreammind_hip_$v.so
  STREAMMIND_HIP_LIB=$L timeout 600 python bench.py $BENCH_FAST ${EXTRA:-} --batch $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']
        print('$v', 'batch $B ${EXTRA:-}', 'frames/s', d['value'], 'gemm_us', r['avg_launch_us'], {k: v['avg_launch_us'] for k, v in r['by_shape'].items()})"
done; done; done | tee gpurun_out/${TAG:-r06k}/nt_ab_${BATCH// /_}${EXTRA:+_fp16}.txt
