"""dev probe: per-round durations of the kinodynamic solve's kernels from a rocprofv3 --kernel-trace run of tools/bench_kd_solve.py --reps 1
    python tools/dev/kd_timeline.py DIR   -> one line per 10 rounds: iteration / Hessian / Jacobian kernel time and the idle gaps between launches"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
kd = [r for r in rows if "landing_kd_" in r[2] or "landing_kinodyn" in r[2]]
names = sorted({r[2][:50] for r in kd}); print(names)
it = [i for i, r in enumerate(kd) if "iter" in r[2]]
print("launches", len(kd), "iteration launches", len(it), "span %.1f ms" % ((kd[-1][1] - kd[0][0]) / 1e6))
prev_end = kd[0][0]; rnd = 0; acc = {}
def flush(lo, hi):
    print("rounds %3d-%3d: " % (lo, hi) + "  ".join("%s %.2f" % (k, v) for k, v in sorted(acc.items())) + "  (ms per round)")
n = 10; start = 0
for r in kd:
    key = "iter" if "iter" in r[2] else "hess" if "hess" in r[2] else "jac" if "jac" in r[2] else "head" if "kd_head" in r[2] else "cond" if "kd_condense" in r[2] else "other"
    acc[key] = acc.get(key, 0) + (r[1] - r[0]) / 1e6 / n
    acc["gap"] = acc.get("gap", 0) + max(0, r[0] - prev_end) / 1e6 / n
    prev_end = max(prev_end, r[1])
    if key == "iter":
        rnd += 1
        if rnd % n == 0:
            flush(rnd - n, rnd - 1); acc = {}
if acc:
    flush(rnd - rnd % n, rnd)
