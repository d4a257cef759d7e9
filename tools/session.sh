#!/bin/bash
# One parametrised command file for the GPU box (replaces the per-call r04_*.sh files): `gpurun -- 'bash tools/session.sh <step> [args]'`.
# Every step writes under gpurun_out/<tag>/ (tag = $TAG or the step name).  Steps can be chained: `bash tools/session.sh a b c`.
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH_FAST="--no-cpu-baseline --no-decode --no-aux --no-e2e --no-fp8"
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d.get('roofline') or {}
        print('$1', 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'gemm_frac', r.get('frac'), 'gemm_us', r.get('avg_launch_us'))"; }
for STEP in "$@"; do
O=gpurun_out/${TAG:-$STEP}; mkdir -p $O
case $STEP in
barrier)      # device-side grid barrier in the price table's forms + the graph-replay boundary slope
  hipcc --offload-arch=gfx950 -O3 -o /tmp/gbx tools/experiments/grid_barrier_xcd_ubench.hip && timeout 300 /tmp/gbx | tee $O/grid_barrier_xcd_ubench.txt ;;
light_tests)  # the tower tests with the light row pass in both modes
  for L in 1 2; do SM_LIGHT=$L timeout 1500 python -m pytest tests/test_gpu_path.py tests/test_gpu_gemm256.py tests/test_gpu_ops.py -q -x -m gpu -k "full_size or lanes or vit or post_ln or gemm256" 2>&1 | tail -3 | sed "s/^/SM_LIGHT=$L /"; done | tee $O/light_tests.txt ;;
light_ab)     # same-box A/B of the default schedule and the single-lane schedule, SM_LIGHT = 0 / 1 / 2, alternating
  for i in 1 2; do for L in 0 1 2; do
    SM_LIGHT=$L python bench.py $BENCH_FAST 2>/dev/null | line "default SM_LIGHT=$L"
    SM_LIGHT=$L python bench.py $BENCH_FAST --batch 28 --no-pipeline 2>/dev/null | line "single-lane SM_LIGHT=$L"
  done; done | tee $O/light_ab.txt ;;
light_trace)  # kernel trace of the default schedule with the light pass: do the light kernels run UNDER the other lane's GEMMs?
  for L in 0 2; do rm -rf /tmp/pl$L; SM_LIGHT=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl$L -- python bench.py --steps 6 --warmup 2 $BENCH_FAST --no-prof > $O/trace_light$L.log 2>&1
    cp "$(find /tmp/pl$L -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_light$L.csv; cp "$(find /tmp/pl$L -name '*kernel_trace.csv' | head -1)" /tmp/kt_light$L.csv
    python tools/trace_overlap.py /tmp/kt_light$L.csv > $O/overlap_light$L.txt 2>&1; head -12 $O/kernel_stats_light$L.csv | cut -c1-150; cat $O/overlap_light$L.txt; done ;;
tests)        timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee $O/pytest_gpu_tail.txt ;;
bench)        timeout 900 python bench.py 2>/dev/null | grep '^{"metric"' > $O/bench_default.json; cut -c1-600 $O/bench_default.json ;;
*) echo "unknown step $STEP" ;;
esac
done
