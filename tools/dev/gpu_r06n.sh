#!/bin/bash
out=gpurun_out/r06n; mkdir -p $out
python tools/soak.py --N 40 --form ccc --batches 16 > $out/soak_ccc.json 2> $out/err.txt
python tools/soak.py --N 40 --form running --batches 16 > $out/soak_running.json 2>> $out/err.txt
python tools/soak.py --N 40 --batches 128 --seed0 500000 > $out/soak_holdout.json 2>> $out/err.txt
python tools/soak.py --N 64 --batches 16 > $out/soak_n64.json 2>> $out/err.txt
: > $out/bench_solve_sizes.jsonl
for b in 64 256 512 1024 2048 4096 8192; do python tools/bench_solve.py --B $b --steps 3 | tail -1 >> $out/bench_solve_sizes.jsonl 2>> $out/err.txt; done
python tools/dev/phase_time.py > $out/phase_time.txt 2>> $out/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06n/soak*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], {k:d[k] for k in ("converged","max_iter_hit","numerical","certified_locally_infeasible","stalled","iters_mean","iters_max","batch_ms_mean")})
for l in open("gpurun_out/r06n/bench_solve_sizes.jsonl"):
    r=json.loads(l); print(r["B"], round(r["sec"]*1e3,1), round(r["nlp_per_s"]), r["converged"])
PY
