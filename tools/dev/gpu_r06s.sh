#!/bin/bash
# KD split build: timeline + batch times
out=$GRAFT_REPO_ROOT/gpurun_out/r06s; mkdir -p $out
cd $GRAFT_REPO_ROOT
python tools/bench_kd_solve.py --inflight 2 > $out/kd_bench.json 2>> $out/err.log
for s in 101 103; do python tools/bench_kd_solve.py --seed $s --reps 2 >> $out/kd_main.jsonl 2>> $out/err.log; done
for s in 100 101 105; do python tools/bench_kd_solve.py --law datagen --seed $s --reps 1 >> $out/kd_dg.jsonl 2>> $out/err.log; done
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06s/kd_bench.json")); print("bench", d["refinement_s"], d["status_counts"], d["iters_max"], d["in_flight"]["s_per_batch"], d["in_flight"]["same_results_as_one_at_a_time"])
for f in ("kd_main","kd_dg"):
    for l in open("gpurun_out/r06s/%s.jsonl"%f):
        d=json.loads(l); print(f, d["what"][-22:], d["refinement_s_best"], d["status_counts"], d["iters_max"])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kdt_new
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kdt_new -- python3 $GRAFT_REPO_ROOT/tools/bench_kd_solve.py --reps 1 > $out/bench_new.json 2> $out/err_new.log
python3 $GRAFT_REPO_ROOT/tools/dev/kd_timeline.py /tmp/kdt_new > $out/timeline_new.txt
cat $out/timeline_new.txt | cut -c1-200
