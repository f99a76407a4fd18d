#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06u; mkdir -p $out
LANDING_KD_PROF=1 LANDING_LIB=landing-controller_amd/_var/lib_kdsdev.so python3 tools/bench_kd_solve.py --reps 1 > $out/bench_dev.json 2> $out/prof.log
grep "kd prof" $out/prof.log
python3 tools/bench_kd_solve.py --reps 3 --inflight 2 > $out/bench.json 2>> $out/err.log
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06u/bench.json")); print(d["refinement_s"], d["status_counts"], d["in_flight"]["s_per_batch"])
PY
