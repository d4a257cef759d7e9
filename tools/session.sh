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
cores)        # co-residency probe (needs tools/experiments/ln_light_coresident.patch applied): light vs ordinary LayerNorm beside a GEMM of another stream
  for L in 1 0 1 0; do SM_NORM_LIGHT=$L timeout 600 python tools/coresidency_probe.py 2>&1 | grep -v Warning; done | tee $O/coresidency_probe.txt ;;
tests)        timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee $O/pytest_gpu_tail.txt ;;
lnx)          # (needs tools/experiments/ln_exchange_epilogue.patch applied) fused LayerNorm in the 256 x 256 epilogue: operator tests, then same-box A/B of the bench (default schedule and single lane)
  timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "post_ln" 2>&1 | tail -5 | tee $O/pytest_post_ln.txt
  for L in 1 0 1 0; do
    SM_GEMM_LNX=$L timeout 600 python bench.py $BENCH_FAST 2>/dev/null | line "default SM_GEMM_LNX=$L"
    SM_GEMM_LNX=$L timeout 600 python bench.py $BENCH_FAST --batch 28 --no-pipeline 2>/dev/null | line "single-lane SM_GEMM_LNX=$L"
  done | tee $O/lnx_ab.txt ;;
lnxtl)        # (needs the same patch) phase timeline of the fused LayerNorm epilogue
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSM_GEMM_TIMELINE -Istreammind_amd/csrc -Iinclude tools/lnx_timeline.hip -o /tmp/lnxtl && for Z in 1 8 32; do echo "SM_LNX_POLL_SLEEPS=$Z"; SM_LNX_POLL_SLEEPS=$Z timeout 300 /tmp/lnxtl; done | tee $O/lnx_timeline.txt ;;
gdtrace)      # kernel trace of the grouped decode step at $GD streams (default 128) -> per-kernel us per step
  rm -rf /tmp/prof_gd; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gd -- python tools/group_decode_bench.py ${GD:-128} > $O/gd_profiled.log 2>&1
  cp "$(find /tmp/prof_gd -name '*kernel_stats.csv' | head -1)" $O/group_decode${GD:-128}_kernel_stats.csv; tail -1 $O/gd_profiled.log
  python tools/trace_window.py /tmp/prof_gd embed_tokens_seg_kernel $(( ${GD:-128} <= 32 ? 1 : (${GD:-128} + 127) / 128 )) > $O/group_decode${GD:-128}_step_timeline.txt; python - $O/group_decode${GD:-128}_step_timeline.txt <<'PY'
import sys, collections, re
d = collections.defaultdict(lambda: [0.0, 0])
for ln in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+) us  dur\s+([\d.]+)  grid\s+(\S+)x\s*(\S+) wg\s+(\S+)  (.*)", ln)
    if m:
        k = m.group(6).split("(")[0].replace("void ", "")[:70]
        d[k][0] += float(m.group(2)); d[k][1] += 1
    else:
        print(ln.strip())
for k, (t, n) in sorted(d.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:72s} n={n:4d} avg {t / n:7.2f} us  total {t:8.1f} us")
PY
  ;;
pftrace)      # kernel trace of a 2048-token prefill (bf16, then fp8 x fp8): tools/prefill_breakdown.py
  rm -rf /tmp/prof_pf; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_pf -- python tools/decode_bench.py 8 4096 1976 > $O/pf_profiled.log 2>&1
  python tools/prefill_breakdown.py /tmp/prof_pf | tee $O/prefill2048_breakdown.txt ;;
dual)         # SwiGLU-dual epilogue + split-K slabs on the 256 tile: operator tests, 2048-token prefill A/B ($KNOB = SM_SWIGLU_FUSE | SM_GEMM256_SPLITK)
  timeout 900 python -m pytest tests/test_gpu_gemm256.py -q -x -k "swiglu or splitk_slabs" 2>&1 | tail -4
  for L in 1 0 1 0; do env ${KNOB:-SM_SWIGLU_FUSE}=$L timeout 600 python tools/decode_bench.py 8 4096 1976 2>&1 | grep -o "prefill_ms[^,]*, .prefill_tokens_per_s[^,]*" | sed "s/^/${KNOB:-SM_SWIGLU_FUSE}=$L /"; done | tee $O/${KNOB:-SM_SWIGLU_FUSE}_ab.txt ;;
graph)        timeout 900 python tools/graph_ab.py 2>/dev/null > $O/graph_ab.json; grep -E '"what"|eager_us"|graph_us"|over_eager' $O/graph_ab.json
  FP8=1 timeout 900 python tools/graph_ab.py 2>/dev/null > $O/graph_ab_fp8.json; grep -E '"what"|eager_us"|graph_us"|over_eager' $O/graph_ab_fp8.json
  timeout 900 python -m pytest tests/test_gpu_graph.py -q 2>&1 | tail -3 ;;
fold)         # round 6: LayerNorm folded into the 256 x 256 products (SM_VIT_LN_FOLD): operator tests, full-size parity probe (bf16 / fp16 tower), same-box A/B of the bench
  timeout 900 python -m pytest tests/test_gpu_gemm256.py -q -x -k "layernorm" 2>&1 | tail -6 | tee $O/pytest_fold_ops.txt
  (timeout 900 python tools/fullsize_parity_probe.py; VIT_FP16=1 timeout 900 python tools/fullsize_parity_probe.py; SM_VIT_LN_FOLD=0 timeout 900 python tools/fullsize_parity_probe.py) 2>&1 | grep -v Warning | tee $O/fold_parity_probe.txt
  for L in 1 0 1 0; do
    SM_VIT_LN_FOLD=$L timeout 600 python bench.py $BENCH_FAST --batch 56 2>/dev/null | line "two-lanes SM_VIT_LN_FOLD=$L"
    SM_VIT_LN_FOLD=$L timeout 600 python bench.py $BENCH_FAST --batch 28 2>/dev/null | line "single-lane-pipelined SM_VIT_LN_FOLD=$L"
  done | tee $O/fold_ab.txt ;;
foldab)       # the A/B alone, bench lines kept (roofline.by_shape: which product pays for the fold in situ)
  for L in 1 0 1 0; do
    SM_VIT_LN_FOLD=$L timeout 600 python bench.py $BENCH_FAST ${EXTRA:-} --batch 56 2>/dev/null | grep '^{"metric"' > $O/bench_two_lanes_fold$L.json; line "two-lanes ${EXTRA:-} SM_VIT_LN_FOLD=$L" < $O/bench_two_lanes_fold$L.json
    SM_VIT_LN_FOLD=$L timeout 600 python bench.py $BENCH_FAST ${EXTRA:-} --batch 28 2>/dev/null | grep '^{"metric"' > $O/bench_single_lane_fold$L.json; line "single-lane-pipelined ${EXTRA:-} SM_VIT_LN_FOLD=$L" < $O/bench_single_lane_fold$L.json
    python -c "
import json
for f in ('two_lanes', 'single_lane'):
    d = json.load(open('$O/bench_%s_fold$L.json' % f))['roofline']['by_shape']
    print('   by_shape', f, {k: v['avg_launch_us'] for k, v in d.items()})"
  done | tee $O/fold_ab${EXTRA:+_fp16}.txt ;;
bench)        timeout 900 python bench.py 2>/dev/null | grep '^{"metric"' > $O/bench_default.json; cut -c1-600 $O/bench_default.json ;;
*) echo "unknown step $STEP" ;;
esac
done
