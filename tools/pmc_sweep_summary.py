"""HBM traffic of the function layer from two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of `python3 tools/bench_sweep.py --B 4096`:
per landing_eval_batch call = the sum over the kernels one call launches (Jacobian / Hessian / g streams of landing_sweep_kernel and the
misc kernel), corrected like tools/pmc_summary.py (FETCH_SIZE x 2, WRITE_SIZE x 1: calibration of the same round's *_pmc_ipm.json).
   python tools/pmc_sweep_summary.py <round-tag> <dir_fetch> <dir_write> <calls>   -> profiles/<tag>_pmc_sweep.json"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def per_kernel(d, name):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "landing_" in k and r["Counter_Name"] == name:
                acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return acc, {k: len(v) for k, v in n.items()}


if __name__ == "__main__":
    tag, d_f, d_w, calls = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    out = {"what": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 tools/bench_sweep.py --B 4096 --steps 20 --warmup 3`: "
                   "HBM bytes per landing_eval_batch call, summed over the kernels one call launches", "kernel_source_sha256": bench.kernel_source_sha(), "members": 4096, "calls": calls}
    cal = None
    p = os.path.join(ROOT, "profiles", "%s_pmc_ipm.json" % tag)
    cf, cw = 2.0, 1.0
    if os.path.exists(p):
        cal = json.load(open(p)).get("calibration") or {}
        g = cal.get("stream8_FETCH_SIZE"); w = cal.get("stream8_WRITE_SIZE")
        if g: cf = 1.0 / g["counter_over_true"]
        if w: cw = 1.0 / w["counter_over_true"]
    fa, fn = per_kernel(d_f, "FETCH_SIZE"); wa, wn = per_kernel(d_w, "WRITE_SIZE")
    kern = {}
    for k in sorted(set(fa) | set(wa)):
        kern[k[:90]] = {"launches_per_call": fn.get(k, wn.get(k, 0)) / calls, "fetch_bytes_per_call": fa.get(k, 0.0) * 1024.0 * cf / calls, "write_bytes_per_call": wa.get(k, 0.0) * 1024.0 * cw / calls}
    out["kernels"] = kern
    out["fetch_bytes_per_call"] = sum(v["fetch_bytes_per_call"] for v in kern.values())
    out["write_bytes_per_call"] = sum(v["write_bytes_per_call"] for v in kern.values())
    out["traffic_bytes_per_call"] = out["fetch_bytes_per_call"] + out["write_bytes_per_call"]
    out["note"] = "KB counters; corrected with the streaming calibration of %s_pmc_ipm.json (factors %.3f read, %.3f write)" % (tag, cf, cw)
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_pmc_sweep.json" % tag), "w"), indent=1)
    print(json.dumps(out, indent=1))
