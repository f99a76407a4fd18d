#!/bin/bash
out=gpurun_out/r06l; mkdir -p $out
run() { label=$1; shift; timeout 200 python tools/bench_kd_solve.py --reps 1 "$@" 2>> $out/err.txt | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.read()); print('$label', round(r['refinement_s_best'],3), r['status_counts'], r['iters_max'])
except Exception as e: print('$label', 'FAILED', e)
"; }
run "intree dflt"
run "intree noclone" --opt kd_clone_after=0
python -m pytest tests/test_gpu_kd_solver.py -x -q 2>&1 | tail -4
grep -v amdgpu $out/err.txt | tail -5
