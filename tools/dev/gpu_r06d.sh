#!/bin/bash
out=gpurun_out/r06d; mkdir -p $out
python tests/make_golden_kd.py $out/n1_kd_solved.npz > $out/golden.log 2>&1
cp $out/n1_kd_solved.npz tests/golden/n1_kd_solved.npz
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
python -m pytest tests/test_kd_solver_cpu.py -q -k gpu_solutions >> $out/pytest.log 2>&1
python tools/bench_kd_solve.py > $out/kd_bench.json 2> $out/kd_bench.err
tail -4 $out/golden.log; tail -6 $out/pytest.log; cut -c1-1500 $out/kd_bench.json
