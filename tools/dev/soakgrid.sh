# Development probe: tools/soak.py for a list of option sets, on 64 fresh batches and on the eight bench batches.
#   bash tools/dev/soakgrid.sh "opt=val,opt=val" ["..." ...]
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%-44s"%d["options"], "conv", d["converged"], "it %.2f p99 %.0f p999 %.0f max %d | ms mean %.1f max %.1f -> %.0f/s"%(d["iters_mean"],d["iters_p99"],d["iters_p999"],d["iters_max"],d["batch_ms_mean"],d["batch_ms_max"],d["nlps_per_s_mean"]))'
for o in "$@"; do
  python tools/soak.py --batches 64 --opts "$o" 2>/dev/null | python -c "$fmt"
  python tools/soak.py --batches 8 --seed0 20211 --seed-step 1000 --opts "$o" 2>/dev/null | python -c "$fmt"
done
