#!/bin/bash
out=gpurun_out/r06f; mkdir -p $out; : > $out/res.txt
run() { # label law seeds opts...
  label=$1; law=$2; seeds=$3; shift 3
  for s in $seeds; do python tools/bench_kd_solve.py --law $law --seed $s --reps 1 "$@" 2>> $out/err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$label', '$law', $s, round(r['refinement_s_best'],3), r['status_counts'], r['iters_max'])" >> $out/res.txt; done
}
run base datagen "100 101 102 103 104 105"
run jam8 datagen "100 101 102 103 104 105" --opt feas_jam=8
run jam12 datagen "100 101 102 103 104 105" --opt feas_jam=12
run jam8max2 datagen "100 101 102 103 104 105" --opt feas_jam=8 --opt feas_max=2
run base main "100 101 102 103"
run jam8 main "100 101 102 103" --opt feas_jam=8
run jam12 main "100 101 102 103" --opt feas_jam=12
cat $out/res.txt
