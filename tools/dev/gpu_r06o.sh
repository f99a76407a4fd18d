#!/bin/bash
# KD, law datagen hold-out seeds: what the plain-iteration limit buys (members that converge between M and 500 iterations against the time of the tail)
out=gpurun_out/r06o; mkdir -p $out
for s in 100 101 102 103 104 105; do
  for m in 500 250 150; do
    python tools/bench_kd_solve.py --law datagen --seed $s --reps 1 --max-iter $m --dump $out/d_${s}_$m.npz >> $out/kd_$m.jsonl 2>> $out/err.log
  done
done
for m in 500 250 150; do
  python tools/bench_kd_solve.py --law main --seed 20211 --reps 1 --max-iter $m --dump $out/m_20211_$m.npz >> $out/kdmain_$m.jsonl 2>> $out/err.log
  python tools/bench_kd_solve.py --law main --seed 103 --reps 1 --max-iter $m --dump $out/m_103_$m.npz >> $out/kdmain_$m.jsonl 2>> $out/err.log
done
python - <<'PY'
import json
for m in (500,250,150):
    for f in ("kd_%d"%m,"kdmain_%d"%m):
        for l in open("gpurun_out/r06o/%s.jsonl"%f):
            d=json.loads(l); print(f, d["what"][-22:], d["refinement_s_best"], d["status_counts"], d["iters_max"])
PY
