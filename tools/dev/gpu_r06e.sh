#!/bin/bash
out=gpurun_out/r06e; mkdir -p $out
: > $out/kd_holdout_datagen.jsonl; : > $out/kd_holdout_main.jsonl
for s in 100 101 102 103 104 105; do python tools/bench_kd_solve.py --law datagen --seed $s --reps 2 >> $out/kd_holdout_datagen.jsonl 2>> $out/err.txt; done
for s in 100 101 102 103 104 105 106 107 108 109 110 111 112 113 114 115; do python tools/bench_kd_solve.py --law main --seed $s --reps 2 >> $out/kd_holdout_main.jsonl 2>> $out/err.txt; done
python tools/bench_kd_solve.py --inflight 2 > $out/kd_bench.json 2>> $out/err.txt
python - <<'PY'
import json
for f in ("gpurun_out/r06e/kd_holdout_datagen.jsonl","gpurun_out/r06e/kd_holdout_main.jsonl"):
    rows=[json.loads(l) for l in open(f) if l.strip()]
    for r in rows: print(r["what"][-30:], round(r["refinement_s_best"],3), r["status_counts"], r["iters_max"])
PY
