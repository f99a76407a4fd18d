#!/bin/bash
out=gpurun_out/r06w; mkdir -p $out
for v in kdd1 kdd2; do
  LANDING_LIB=landing-controller_amd/_var/lib_$v.so python tools/bench_kd_solve.py --reps 3 > $out/main_$v.json 2>> $out/err.log
  LANDING_LIB=landing-controller_amd/_var/lib_$v.so python tools/bench_kd_solve.py --law datagen --seed 101 --reps 1 > $out/dg_$v.json 2>> $out/err.log
done
python - <<'PY'
import json
for v in ("kdd1","kdd2"):
    for f in ("main","dg"):
        d=json.load(open("gpurun_out/r06w/%s_%s.json"%(f,v))); print(v, f, d["refinement_s_best"], d["status_counts"], d["iters_max"])
PY
