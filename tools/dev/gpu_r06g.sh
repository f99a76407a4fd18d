#!/bin/bash
out=gpurun_out/r06g; mkdir -p $out; : > $out/res.txt
run() { label=$1; law=$2; seeds=$3; shift 3
  for s in $seeds; do python tools/bench_kd_solve.py --law $law --seed $s --reps 1 "$@" 2>> $out/err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$label', '$law', $s, round(r['refinement_s_best'],3), r['status_counts'], r['iters_max'])" >> $out/res.txt; done
}
run dflt datagen "100 101 102 103 104 105"
run dec0 datagen "100 101 102 103 104 105" --opt feas_delta_dec=0
run dflt main "100 101"
cat $out/res.txt
