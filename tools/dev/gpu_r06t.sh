#!/bin/bash
# KD condense kernel variants: batch time + the kernel's share (rocprofv3 stats)
out=$GRAFT_REPO_ROOT/gpurun_out/r06t; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in a b c d e f; do
  export LANDING_LIB=$GRAFT_REPO_ROOT/landing-controller_amd/_var/lib_kds_$v.so
  python3 $GRAFT_REPO_ROOT/tools/bench_kd_solve.py --reps 3 > $out/bench_$v.json 2>> $out/err.log
  rm -rf /tmp/kdt_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kdt_$v -- python3 $GRAFT_REPO_ROOT/tools/bench_kd_solve.py --reps 1 > /dev/null 2>> $out/err.log
  f=$(find /tmp/kdt_$v -name "*kernel_stats.csv" | head -1)
  python3 - $v $f <<'PY'
import sys, csv, json
v, f = sys.argv[1], sys.argv[2]
d = json.load(open("/root/repo/gpurun_out/r06t/bench_%s.json" % v))
rows = {r["Name"][:40]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
print(v, d["refinement_s"], d["status_counts"], {k.split("landing::")[-1][:28]: round(t, 1) for k, (c, t) in rows.items() if "kd_" in k or "kinodyn" in k})
PY
done
