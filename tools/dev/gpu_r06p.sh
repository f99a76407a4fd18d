#!/bin/bash
# KD: what the chunk loop of the condensation (J' Sigma J on the matrix cores, staged through LDS) costs -- builds that run it 1 / 2 / 3 times (same results)
out=gpurun_out/r06p; mkdir -p $out
for v in kdc1 kdc2 kdc3; do
  LANDING_LIB=landing-controller_amd/_var/lib_$v.so python tools/bench_kd_solve.py --reps 2 > $out/main_$v.json 2>> $out/err.log
  LANDING_LIB=landing-controller_amd/_var/lib_$v.so python tools/bench_kd_solve.py --law datagen --seed 101 --reps 1 > $out/dg_$v.json 2>> $out/err.log
done
python - <<'PY'
import json
for v in ("kdc1","kdc2","kdc3"):
    for f in ("main","dg"):
        d=json.load(open("gpurun_out/r06p/%s_%s.json"%(f,v))); print(v, f, d["refinement_s_best"], d["status_counts"], d["iters_max"], d.get("rounds"))
PY
