#!/bin/bash
out=gpurun_out/r06b; mkdir -p $out
python tools/dev/phase_time.py > $out/phase_time.txt 2>&1
for law in datagen main; do
  python tools/soak.py --N 20 --grid reference --law $law --batches 16 > $out/soak_n20_$law.json 2> $out/err.txt
  python tools/soak.py --N 20 --grid reference --law $law --batches 16 --opts feas_max=2 > $out/soak_n20_${law}_max2.json 2>> $out/err.txt
  python tools/soak.py --N 20 --grid reference --law $law --batches 16 --opts feas_jam=5 > $out/soak_n20_${law}_jam5.json 2>> $out/err.txt
done
grep -v amdgpu.ids $out/phase_time.txt | cut -c1-200; for f in $out/soak*.json; do echo $f; python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print({k:d[k] for k in ("converged","max_iter_hit","numerical","certified_locally_infeasible","stalled","iters_mean","iters_p999","iters_max","batch_ms_mean","batch_ms_max")})
PY
done
