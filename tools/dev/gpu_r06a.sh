#!/bin/bash
# round 6, first GPU pass: GPU test suite, soaks of every family with the new phase rules, bench line
out=gpurun_out/r06a; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
python tools/soak.py --N 20 --grid reference --law datagen --batches 16 > $out/soak_n20_datagen.json 2> $out/soak_n20_datagen.err
python tools/soak.py --N 20 --grid reference --law main --batches 16 > $out/soak_n20_main.json 2> $out/soak_n20_main.err
python tools/soak.py --N 40 --form ccc --batches 16 > $out/soak_ccc.json 2> $out/soak_ccc.err
python tools/soak.py --N 40 --form running --batches 16 > $out/soak_running.json 2> $out/soak_running.err
python tools/soak.py --N 40 --batches 128 --seed0 500000 > $out/soak_holdout.json 2> $out/soak_holdout.err
python tools/soak.py --N 64 --batches 16 > $out/soak_n64.json 2> $out/soak_n64.err
python bench.py > $out/bench.json 2> $out/bench.err
tail -3 $out/pytest.log; cat $out/soak_n20_datagen.json | cut -c1-900; cat $out/soak_n20_main.json | cut -c1-700; cut -c1-600 $out/soak_ccc.json; cut -c1-500 $out/soak_holdout.json; cut -c1-1200 $out/bench.json
