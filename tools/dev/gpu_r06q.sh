#!/bin/bash
# KD with the condensation in its own launch: GPU tests of the kinodynamic rows, batch times (law main bench seed + hold-out, law datagen hold-out)
out=gpurun_out/r06q; mkdir -p $out
python -m pytest tests/test_gpu_kd_solver.py tests/test_n1_rows.py tests/test_gpu_sweep.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
python tools/bench_kd_solve.py --inflight 2 > $out/kd_bench.json 2>> $out/err.log
for s in 100 101 102 103; do python tools/bench_kd_solve.py --seed $s --reps 2 >> $out/kd_main.jsonl 2>> $out/err.log; done
for s in 100 101 102 103 104 105; do python tools/bench_kd_solve.py --law datagen --seed $s --reps 1 >> $out/kd_dg.jsonl 2>> $out/err.log; done
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06q/kd_bench.json")); print("bench", d["refinement_s"], d["status_counts"], d["iters_max"], d["in_flight"]["s_per_batch"], d["in_flight"]["same_results_as_one_at_a_time"])
for f in ("kd_main","kd_dg"):
    for l in open("gpurun_out/r06q/%s.jsonl"%f):
        d=json.loads(l); print(f, d["what"][-22:], d["refinement_s_best"], d["status_counts"], d["iters_max"])
PY
