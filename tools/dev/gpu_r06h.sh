#!/bin/bash
out=gpurun_out/r06h; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -4 $out/pytest.log
bash tools/profile_round.sh r06 > $out/profile.log 2>&1
tail -5 $out/profile.log
python bench.py > $out/bench.json 2> $out/bench.err
python tools/bench_solve.py > $out/bench_solve_sizes.jsonl 2>> $out/bench.err
python tools/bench_kd_solve.py --inflight 2 > $out/kd_bench.json 2>> $out/bench.err
python tools/bench_mpc.py > $out/mpc.json 2>> $out/bench.err
python tools/dev/itdump.py r06 > $out/itdump.log 2>> $out/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06h/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["pcie_inclusive"]["value"], d["two_batches_in_flight"]["value"], d["streamed"]["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["sweep"]["frac"], d["cpu_baseline"]["value"], d["next_rows"])
PY
