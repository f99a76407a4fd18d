#!/bin/bash
out=gpurun_out/r06j; mkdir -p $out
run() { label=$1; lib=$2; shift 2; LANDING_LIB=$lib timeout 200 python tools/bench_kd_solve.py --reps 1 "$@" 2>> $out/err.txt | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.read()); print('$label', round(r['refinement_s_best'],3), r['status_counts'], r['iters_max'])
except Exception as e: print('$label', 'FAILED', e)
"; }
for c in 59a2bd9 7cc32bf; do
  run "$c dflt" $PWD/tools/dev/libkd_$c.so
  run "$c noclone" $PWD/tools/dev/libkd_$c.so --opt kd_clone_after=0
done
run "HEAD dflt" $PWD/landing-controller_amd/liblanding_mi355x.so
grep -v amdgpu $out/err.txt | tail -5
