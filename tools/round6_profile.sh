#!/bin/bash
# Round-6 end-of-round measurement set, ONE gpurun call on the FINAL revision (the round-5 verdict: the r05 kernel stats / PMC files were six hours and ~35
# commits older than the tree they were committed with).  Everything lands in gpurun_out/round6/; copy to profiles/r06_*.
#   gpurun --timeout 3000 -- 'bash tools/round6_profile.sh'
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round6; mkdir -p $O
git -C . rev-parse HEAD > $O/revision.txt 2>/dev/null || sha256sum bench.py streammind_amd/lib/libstreammind_hip.so | cut -c1-16 > $O/revision.txt
STEPS="--no-cpu-baseline --no-decode --no-aux --no-e2e --no-prof"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | tee $O/pytest_gpu_tail.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/pytest_gpu_tail.txt
fi
timeout 900 python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log > $O/bench_default.json; cut -c1-300 $O/bench_default.json
# kernel trace of the bench STEPS, single-lane plain schedule (the schedule of the line's `roofline` segment): default path (bf16: LayerNorm launches) ...
rm -rf /tmp/p1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python bench.py --batch 28 --no-pipeline --steps 20 --warmup 2 $STEPS > $O/bench_steps_profiled.log 2>&1
grep '^{"metric"' $O/bench_steps_profiled.log > $O/bench_steps_profiled.json; cp "$(find /tmp/p1 -name '*kernel_stats.csv' | head -1)" $O/bench_steps_kernel_stats.csv
# ... the same with the bf16 fold forced on, and the fp16 tower (folds by default)
rm -rf /tmp/p2; SM_VIT_LN_FOLD=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python bench.py --batch 28 --no-pipeline --steps 20 --warmup 2 $STEPS > /dev/null 2>&1
cp "$(find /tmp/p2 -name '*kernel_stats.csv' | head -1)" $O/bench_steps_fold_kernel_stats.csv
rm -rf /tmp/p3; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python bench.py --vit-fp16 --batch 28 --no-pipeline --steps 20 --warmup 2 $STEPS > /dev/null 2>&1
cp "$(find /tmp/p3 -name '*kernel_stats.csv' | head -1)" $O/bench_steps_fp16_fold_kernel_stats.csv
# ... and the default schedule as the line times it (two kernels share the chip: a throughput schedule, not kernel times)
rm -rf /tmp/p4; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -- python bench.py --batch 56 --steps 10 --warmup 2 $STEPS > /dev/null 2>&1
cp "$(find /tmp/p4 -name '*kernel_stats.csv' | head -1)" $O/bench_steps_two_lanes_kernel_stats.csv
# PMC passes, each counter group in its own run, kernel trace only (no sys / hip trace): fabric bytes and MFMA busy of the single-lane steps
PMCARGS="--batch 28 --no-pipeline --steps 3 --warmup 1 --stream-frames 112 --no-cpu-baseline --no-decode --no-prof --no-aux --no-e2e"
rm -rf /tmp/pf /tmp/pw /tmp/pm
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python bench.py $PMCARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python bench.py $PMCARGS > /dev/null 2>&1
python tools/pmc_traffic_summary.py /tmp/pf /tmp/pw $O/gemm_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d /tmp/pm -- python bench.py $PMCARGS > /dev/null 2>&1
python tools/pmc_mfma_summary.py /tmp/pm $O/mfma_util.json
rm -rf /tmp/pm16; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d /tmp/pm16 -- python bench.py --vit-fp16 $PMCARGS > /dev/null 2>&1
python tools/pmc_mfma_summary.py /tmp/pm16 $O/mfma_util_fp16_fold.json
# decode (bf16 / fp8), per-call latency, the pass
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -- python tools/decode_bench.py 64 1024 > /dev/null 2>&1; cp "$(find /tmp/pd -name '*kernel_stats.csv' | head -1)" $O/decode_kernel_stats.csv
for F in 1 4 7 8 10 14 16 17 20; do timeout 300 python tools/tick_bench.py $F 100 2>&1 | tail -1; done > $O/tick_latency.txt; cat $O/tick_latency.txt
for R in 1 28; do timeout 300 python tools/pass_bench.py $R 200 2>&1 | tail -1; done > $O/pass_bench.txt; cat $O/pass_bench.txt
timeout 300 python tools/fold_bench.py 28 2>&1 | grep -v Warning > $O/fold_bench.txt; cat $O/fold_bench.txt
# LLM side: per-kernel time of a 2048-token and of a 328-token prefill, the step timeline of a 128- and a 512-stream batched decode, the prefill attention in isolation
for N in 1976 256; do rm -rf /tmp/prof_pf; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_pf -- python tools/decode_bench.py 8 4096 $N > /dev/null 2>&1
  python tools/prefill_breakdown.py /tmp/prof_pf > $O/prefill$((N + 72))_breakdown.txt; head -8 $O/prefill$((N + 72))_breakdown.txt; done
for GD in 128 512; do rm -rf /tmp/prof_gd; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gd -- python tools/group_decode_bench.py $GD > /dev/null 2>&1
  python tools/trace_window.py /tmp/prof_gd embed_tokens_seg_kernel $(( (GD + 127) / 128 )) > $O/group_decode${GD}_step_timeline.txt; head -12 $O/group_decode${GD}_step_timeline.txt | cut -c1-150; done
for shape in "2048 0" "512 0" "2048 2048" "300 0"; do for P in 1 0; do SM_ATTN_PREFILL=$P timeout 120 python tools/attn_causal_bench.py $shape 2>/dev/null; done; done > $O/attn_causal_bench.txt; cat $O/attn_causal_bench.txt
head -8 $O/bench_steps_kernel_stats.csv | cut -c1-150; head -6 $O/bench_steps_fp16_fold_kernel_stats.csv | cut -c1-150
