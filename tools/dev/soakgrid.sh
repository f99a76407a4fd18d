fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%-44s"%d["options"], "conv", d["converged"], "it %.2f p999 %.0f max %d | ms mean %.1f max %.1f -> %.0f/s"%(d["iters_mean"],d["iters_p999"],d["iters_max"],d["batch_ms_mean"],d["batch_ms_max"],d["nlps_per_s_mean"]))'
for o in "clip_until=0.05" "clip_until=0.1" "clip_until=0.2" "clip_until=0.3" "clip_until=0.1,restart_period=70" "clip_until=0.2,restart_period=70" "clip_until=0.1,restart_period=75"; do
  python tools/soak.py --batches 64 --opts $o 2>/dev/null | python -c "$fmt"
  python tools/soak.py --batches 8 --seed0 20211 --seed-step 1000 --opts $o 2>/dev/null | python -c "$fmt"
done
