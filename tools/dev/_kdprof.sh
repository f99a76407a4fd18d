mkdir -p gpurun_out/kdprof
timeout 300 python3 tools/bench_kd_solve.py --reps 3 > gpurun_out/kdprof/kd_bench.json 2> gpurun_out/kdprof/kd_bench.err
for seed in 8 9 10 11 12 13 14; do timeout 300 python3 tools/bench_kd_solve.py --reps 2 --seed $seed >> gpurun_out/kdprof/kd_seeds.jsonl 2>/dev/null; done
for seed in 20211 8 9; do timeout 300 python3 tools/bench_kd_solve.py --reps 2 --seed $seed --law datagen >> gpurun_out/kdprof/kd_seeds_datagen.jsonl 2>/dev/null; done
for seed in 20211 8 9 10 11; do timeout 300 python3 tools/bench_kd_solve.py --reps 2 --seed $seed --opt kd_clone_after=0 --opt clip_k=4 --opt restart_period=75 >> gpurun_out/kdprof/kd_seeds_noportfolio.jsonl 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kdprof/stats -- python3 tools/bench_kd_solve.py --reps 1 > gpurun_out/kdprof/stats.json 2> gpurun_out/kdprof/stats.err
python3 tools/dev/kd_timeline.py gpurun_out/kdprof/stats > gpurun_out/kdprof/kd_timeline.txt 2>&1
find gpurun_out/kdprof/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/kdprof/kd_kernel_stats.csv \;
rm -rf gpurun_out/kdprof/stats
