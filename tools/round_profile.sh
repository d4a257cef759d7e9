#!/bin/bash
# End-of-round measurement set (run on the GPU box via gpurun): full GPU test suite, default bench line, rocprofv3 kernel
# stats of the same command, PMC traffic passes.  Summaries land in gpurun_out/round/ -- copy what is to be judged to profiles/.
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round; mkdir -p $O; export TAG=round
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log > $O/bench_default.json; cut -c1-400 $O/bench_default.json
# kernel stats of the BENCH STEPS ONLY (no decode / end-to-end / auxiliary legs, no HIP-event bracketing).  Two runs:
#  (a) the default schedule (56 frames per step, two tower lanes + pipelined gate pass): two kernels share the chip, so a kernel's
#      duration here includes the time its blocks waited for CUs the other lane's kernel held;
#  (b) --batch 28 --no-pipeline (one lane, plain call = the schedule bench.py's `roofline` segment runs): the dominant kernel's
#      undisturbed average can be read straight from this file and must agree with roofline.avg_launch_us
rm -rf /tmp/prof_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-aux --no-e2e --no-prof > $O/bench_profiled.log 2>&1
grep '^{"metric"' $O/bench_profiled.log > $O/bench_profiled.json
cp "$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_steps_default.csv
rm -rf /tmp/prof_stats1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats1 -- python bench.py --batch 28 --no-pipeline --steps 20 --warmup 2 --no-cpu-baseline --no-decode --no-aux --no-e2e --no-prof > $O/bench_profiled_single_lane.log 2>&1
grep '^{"metric"' $O/bench_profiled_single_lane.log > $O/bench_profiled_single_lane.json
cp "$(find /tmp/prof_stats1 -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_steps.csv
# the decode step on its own (Mistral-7B, 64 tokens after a 328-token prefill)
rm -rf /tmp/prof_dec; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python tools/decode_bench.py 64 1024 > $O/decode_profiled.log 2>&1
cp "$(find /tmp/prof_dec -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_decode.csv
# ... and with fp8 weights (weight-only mode, FP8=1)
rm -rf /tmp/prof_dec8; FP8=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec8 -- python tools/decode_bench.py 64 1024 > $O/decode_fp8_profiled.log 2>&1
cp "$(find /tmp/prof_dec8 -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_decode_fp8.csv
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python bench.py --batch 28 --no-pipeline --steps 3 --warmup 1 --stream-frames 112 --no-cpu-baseline --no-decode --no-prof --no-aux > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python bench.py --batch 28 --no-pipeline --steps 3 --warmup 1 --stream-frames 112 --no-cpu-baseline --no-decode --no-prof --no-aux > /dev/null 2>&1
python tools/pmc_traffic_summary.py /tmp/pmc_f /tmp/pmc_w $O/gemm_traffic.json
rm -rf /tmp/pmc_m
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d /tmp/pmc_m -- python bench.py --batch 28 --no-pipeline --steps 3 --warmup 1 --stream-frames 112 --no-cpu-baseline --no-decode --no-prof --no-aux > /dev/null 2>&1
python tools/pmc_mfma_summary.py /tmp/pmc_m $O/mfma_util.json
# round 4: the one-frame tick (kernel stats + one tick's launch timeline), the connector + gate pass alone, tick latencies
for F in 1 4 8; do timeout 300 python tools/tick_bench.py $F 200 2>&1 | tail -1; done > $O/tick_latency.txt; cat $O/tick_latency.txt
for R in 1 28; do timeout 300 python tools/pass_bench.py $R 200 2>&1 | tail -1; done > $O/pass_bench.txt; cat $O/pass_bench.txt
rm -rf /tmp/prof_t1; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t1 -- python tools/tick_bench.py 1 200 > $O/tick_b1_profiled.log 2>&1
cp "$(find /tmp/prof_t1 -name '*kernel_stats.csv' | head -1)" $O/tick_b1_kernel_stats.csv
python tools/trace_window.py /tmp/prof_t1 preprocess_kernel > $O/tick_b1_timeline.txt 2>&1
rm -rf /tmp/prof_p28; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_p28 -- python tools/pass_bench.py 28 40 > $O/pass28_profiled.log 2>&1
python tools/trace_window.py /tmp/prof_p28 gate_tail_kernel > $O/pass28_timeline.txt 2>&1
head -8 $O/kernel_stats_steps_default.csv | cut -c1-160; head -12 $O/kernel_stats_steps.csv | cut -c1-160; head -8 $O/kernel_stats_decode.csv | cut -c1-160
# round 5: the LLM side -- 2048-token prefill (kernel stats + per-kernel breakdown), the batched decode step at 128 and 512 streams (one-step timelines),
# and the tile-walk A/B of the dominant GEMM's fabric traffic (SM_GEMM_CG = column groups per XCD band: 0 = plain row-major walk)
rm -rf /tmp/prof_pf; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pf -- python tools/decode_bench.py 8 4096 1976 > $O/pf_profiled.log 2>&1
cp "$(find /tmp/prof_pf -name '*kernel_stats.csv' | head -1)" $O/prefill_kernel_stats.csv
python tools/prefill_breakdown.py /tmp/prof_pf > $O/prefill2048_breakdown.txt; cat $O/prefill2048_breakdown.txt
for GD in 128 512; do TAG=round GD=$GD bash tools/session.sh gdtrace > /dev/null 2>&1; tail -12 $O/group_decode${GD}_step_timeline.txt | head -0; done
for CG in 0 3; do
  rm -rf /tmp/pmc_f$CG; SM_GEMM_CG=$CG rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f$CG -- python bench.py --batch 28 --no-pipeline --steps 3 --warmup 1 --stream-frames 112 --no-cpu-baseline --no-decode --no-prof --no-aux > /dev/null 2>&1
  rm -rf /tmp/pmc_w$CG; SM_GEMM_CG=$CG rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w$CG -- python bench.py --batch 28 --no-pipeline --steps 3 --warmup 1 --stream-frames 112 --no-cpu-baseline --no-decode --no-prof --no-aux > /dev/null 2>&1
  python tools/pmc_traffic_summary.py /tmp/pmc_f$CG /tmp/pmc_w$CG $O/gemm_traffic_cg$CG.json
  SM_GEMM_CG=$CG timeout 600 python bench.py --batch 28 --no-pipeline --no-cpu-baseline --no-decode --no-aux --no-e2e --no-fp8 2>/dev/null | grep '^{"metric"' > $O/bench_single_lane_cg$CG.json
done
python - <<'PY'
import json
for cg in (0, 3):
    t = json.load(open(f"gpurun_out/round/gemm_traffic_cg{cg}.json")); b = json.loads(open(f"gpurun_out/round/bench_single_lane_cg{cg}.json").read())
    print(f"SM_GEMM_CG={cg}: hbm bytes per GEMM launch {t.get('hbm_bytes_per_launch')}, single-lane frames/s {b['value']}, gemm avg launch us {b['roofline']['avg_launch_us']}")
PY
