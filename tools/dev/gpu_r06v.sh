#!/bin/bash
# full GPU suite + KD records on the split build
out=gpurun_out/r06v; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
python tools/bench_kd_solve.py --inflight 2 > $out/kd_bench.json 2>> $out/err.log
for s in 100 101 102 103 104 105 106 107 108 109 110 111 112 113 114 115; do python tools/bench_kd_solve.py --seed $s --reps 2 >> $out/kd_holdout_main.jsonl 2>> $out/err.log; done
for s in 100 101 102 103 104 105; do python tools/bench_kd_solve.py --law datagen --seed $s --reps 2 >> $out/kd_holdout_datagen.jsonl 2>> $out/err.log; done
python - <<'PY'
import json, numpy as np
d=json.load(open("gpurun_out/r06v/kd_bench.json")); print("bench", d["refinement_s"], d["status_counts"], d["iters_max"], d["in_flight"]["s_per_batch"], d["in_flight"]["same_results_as_one_at_a_time"])
for f in ("kd_holdout_main","kd_holdout_datagen"):
    t=[]; und=0
    for l in open("gpurun_out/r06v/%s.jsonl"%f):
        d=json.loads(l); t.append(d["refinement_s_best"]); und+=d["status_counts"][1]+d["status_counts"][2]
    print(f, "min %.3f mean %.3f max %.3f undecided %d" % (min(t), np.mean(t), max(t), und), [round(x,3) for x in t])
PY
