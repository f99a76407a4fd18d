mkdir -p gpurun_out/kdexp10
run() { name=$1; shift 1; timeout 300 python tools/bench_kd_solve.py --reps 1 "$@" > gpurun_out/kdexp10/$name.json 2>&1; }
for seed in 20211 9; do
run base_$seed --seed $seed
run bp03_$seed --seed $seed --opt bound_push=0.03 --opt bound_frac=0.03
run bp10_$seed --seed $seed --opt bound_push=0.1 --opt bound_frac=0.1
run mu03_$seed --seed $seed --opt mu_init=0.3
run mu1_$seed --seed $seed --opt mu_init=1.0
run mu003_$seed --seed $seed --opt mu_init=0.03
run ke40_$seed --seed $seed --opt kappa_eps=40
run ke160_$seed --seed $seed --opt kappa_eps=160
run cu01_$seed --seed $seed --opt clip_until=0.01
run cu1_$seed --seed $seed --opt clip_until=0.1
run tm12_$seed --seed $seed --opt theta_mu=1.8
run c24_$seed --seed $seed --opt clip_k=24
done
