#!/bin/bash
# finish kernel: a clone's certificate decides an undecided family -- KD GPU tests + law datagen hold-out outcomes
out=gpurun_out/r06x; mkdir -p $out
python -m pytest tests/test_gpu_kd_solver.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for s in 100 101 102 103 104 105; do python tools/bench_kd_solve.py --law datagen --seed $s --reps 1 >> $out/kd_dg.jsonl 2>> $out/err.log; done
python tools/bench_kd_solve.py --reps 2 > $out/kd_bench.json 2>> $out/err.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06x/kd_bench.json")); print("bench", d["refinement_s"], d["status_counts"])
for l in open("gpurun_out/r06x/kd_dg.jsonl"):
    d=json.loads(l); print(d["what"][-22:], d["refinement_s_best"], d["status_counts"], d["iters_max"])
PY
