#!/bin/bash
out=gpurun_out/r06m; mkdir -p $out
for law in datagen main; do
  python tools/soak.py --N 20 --grid reference --law $law --batches 16 --seed0 300000 > $out/soak_n20_${law}_holdout.json 2>> $out/err.txt
  python tools/soak.py --N 20 --grid reference --law $law --batches 16 > $out/soak_n20_${law}.json 2>> $out/err.txt
done
: > $out/kd_holdout_datagen.jsonl; : > $out/kd_holdout_main.jsonl
for s in 100 101 102 103 104 105; do python tools/bench_kd_solve.py --law datagen --seed $s --reps 2 >> $out/kd_holdout_datagen.jsonl 2>> $out/err.txt; done
for s in 100 101 102 103 104 105 106 107 108 109 110 111 112 113 114 115; do python tools/bench_kd_solve.py --law main --seed $s --reps 2 >> $out/kd_holdout_main.jsonl 2>> $out/err.txt; done
python bench.py > $out/bench.json 2>> $out/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06m/soak*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], {k:d[k] for k in ("converged","max_iter_hit","numerical","certified_locally_infeasible","stalled","iters_p999","iters_max","batch_ms_mean")})
for f in ("gpurun_out/r06m/kd_holdout_datagen.jsonl","gpurun_out/r06m/kd_holdout_main.jsonl"):
    rows=[json.loads(l) for l in open(f) if l.strip()]
    print(f.split("/")[-1], [(round(r["refinement_s_best"],3), r["status_counts"], r["iters_max"]) for r in rows])
d=json.load(open("gpurun_out/r06m/bench.json")); print(d["value"], d["streamed"]["value"], d["two_batches_in_flight"]["value"], d["pcie_inclusive"]["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["sweep"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
