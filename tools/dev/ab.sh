#!/bin/bash
# dev probe (run on the GPU box through gpurun): A/B of library builds -- batch times + phase timers (tools/dev/ab_one.py) and, with PMC=1,
# HBM traffic of landing_ipm_kernel per launch (separate rocprofv3 --pmc passes, FETCH_SIZE x2 + WRITE_SIZE, KB counters).
#   tools/dev/ab.sh OUT lib1.so lib2.so ...       (paths relative to the repo root; "cur" = the product library)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
for lib in "$@"; do
  tag=$(basename $lib .so)
  python3 $GRAFT_REPO_ROOT/tools/dev/ab_one.py $lib ${NSEED:-4} > $out/ab_$tag.json 2> $out/ab_$tag.err
  tail -1 $out/ab_$tag.json
  if [ "$PMC" = "1" ]; then
    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_$c
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $GRAFT_REPO_ROOT/tools/dev/ab_one.py $lib 1 --noprof > /dev/null 2> $out/pmc_${tag}_$c.err
    done
    python3 - $tag <<'PY' | tee -a $out/ab_$tag.traffic
import csv, glob, sys, collections
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = 0.0; n = set()
    for f in glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "landing_ipm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                acc += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    tot[c] = acc / max(len(n), 1) * 1024.0
print("%s traffic per launch: fetch(x2) %.1f GB + write %.1f GB = %.1f GB" % (sys.argv[1], 2 * tot["FETCH_SIZE"] / 1e9, tot["WRITE_SIZE"] / 1e9, (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / 1e9))
PY
    cd $GRAFT_REPO_ROOT
  fi
done
