#!/bin/bash
# Profiles of one round (run on the GPU box through gpurun): kernel-trace stats of bench.py + separate PMC passes
# (never combined with tracing domains other than --kernel-trace).  Output: gpurun_out/prof_$1/*
tag=${1:-r06}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B > $out/bench_stats.json 2> $out/stats.err
timeout -k 5 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B > /dev/null 2> $out/fetch.err
timeout -k 5 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $B > /dev/null 2> $out/write.err
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sq -- $B > /dev/null 2> $out/sq.err
C="python3 $GRAFT_REPO_ROOT/tools/pmc_calib.py"
timeout -k 5 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- $C > /dev/null 2> $out/cal_fetch.err
timeout -k 5 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- $C > /dev/null 2> $out/cal_write.err
S="python3 $GRAFT_REPO_ROOT/tools/bench_sweep.py --B 4096 --steps 20 --warmup 3"      # function layer: 23 landing_eval_batch calls
timeout -k 5 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/sw_fetch -- $S > /dev/null 2> $out/sw_fetch.err
timeout -k 5 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/sw_write -- $S > /dev/null 2> $out/sw_write.err
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $tag $out/stats $out/fetch $out/write $out/sq $out/cal_fetch $out/cal_write > $out/summary.json 2> $out/summary.err
python3 tools/pmc_sweep_summary.py $tag $out/sw_fetch $out/sw_write 23 > $out/summary_sweep.json 2> $out/summary_sweep.err
cp profiles/${tag}_pmc_ipm.json profiles/${tag}_pmc_sweep.json $out/ 2>/dev/null
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
du -sh $out; tail -3 $out/*.err | tail -20
