#!/bin/bash
# KD timeline: split build (product) against the fused build (lib_kdc1 = the round-5 form)
out=$GRAFT_REPO_ROOT/gpurun_out/r06r; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  rm -rf /tmp/kdt_$v
  if [ $v = old ]; then export LANDING_LIB=$GRAFT_REPO_ROOT/landing-controller_amd/_var/lib_kdc1.so; else unset LANDING_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kdt_$v -- python3 $GRAFT_REPO_ROOT/tools/bench_kd_solve.py --reps 1 > $out/bench_$v.json 2> $out/err_$v.log
  python3 $GRAFT_REPO_ROOT/tools/dev/kd_timeline.py /tmp/kdt_$v > $out/timeline_$v.txt
  f=$(find /tmp/kdt_$v -name "*kernel_stats.csv" | head -1); head -12 $f > $out/stats_$v.csv
  cat $out/timeline_$v.txt | cut -c1-200
done
