#!/bin/bash
out=gpurun_out/r06i; mkdir -p $out
run() { label=$1; shift; python tools/bench_kd_solve.py --reps 1 --dump $out/$label.npz "$@" 2>> $out/err.txt | python -c "
import json,sys,numpy as np
r=json.loads(sys.stdin.read()); print('$label', round(r['refinement_s_best'],3), r['status_counts'], r['iters_max'])
d=np.load('$out/$label.npz'); st=d[d.files[0]] if False else None
print('   files', d.files)
s=d['status']; it=d['iters']
print('   slow', [(int(m), int(it[m]), int(s[m])) for m in np.argsort(-it)[:6]], ' cert with iters>0:', [(int(m), int(it[m])) for m in np.nonzero((s==3)&(it>0))[0]][:8])
"; }
run dflt
run noresume --opt feas_resume=0
run noret --opt feas_ret_push=0
run nodec --opt feas_delta_dec=0
run nopolish --opt feas_polish=0
run noclone --opt kd_clone_after=0
cat $out/err.txt | grep -v amdgpu | tail -5
