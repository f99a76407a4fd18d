#!/bin/bash
# rocprofv3 summary of one kinodynamic refinement batch (final build) + the development build's phase timers
out=$GRAFT_REPO_ROOT/gpurun_out/r06y; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kdt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kdt -- python3 $GRAFT_REPO_ROOT/tools/bench_kd_solve.py --reps 1 > $out/bench.json 2> $out/err.log
f=$(find /tmp/kdt -name "*kernel_stats.csv" | head -1); cp $f $out/kd_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/dev/kd_timeline.py /tmp/kdt > $out/kd_timeline.txt
head -12 $out/kd_kernel_stats.csv | cut -c1-150
cd $GRAFT_REPO_ROOT
LANDING_KD_PROF=1 LANDING_LIB=landing-controller_amd/_var/lib_kdsdev.so python3 tools/bench_kd_solve.py --reps 1 > $out/bench_dev.json 2> $out/prof.log
grep "kd prof" $out/prof.log
