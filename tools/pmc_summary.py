"""Reduce rocprofv3 output directories (kernel-trace stats + separate --pmc passes) to the JSON files under profiles/.
   python tools/pmc_summary.py <round-tag> <dir_stats> <dir_fetch> <dir_write> <dir_sq> [<dir_calib_fetch> <dir_calib_write>]"""
import collections, csv, glob, json, os, sys


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return acc, {k: len(v) for k, v in n.items()}


def pick(acc, sub):
    for k in acc:
        if sub in k:
            return k
    return None


def stats(d, sub):
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Name"]:
                return {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "total_ms": float(r["TotalDurationNs"]) / 1e6}
    return None


if __name__ == "__main__":
    tag, d_stats, d_f, d_w, d_sq = sys.argv[1:6]
    out = {"what": "rocprofv3 --kernel-trace --stats and separate --pmc passes of `python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras` (per-launch averages of landing_ipm_kernel)"}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["kernel_source_sha256"] = bench.kernel_source_sha()      # bench.py refuses the traffic figure on any other build of the kernel
    out["kernel_stats"] = stats(d_stats, "landing_ipm_kernel")
    per = {}
    for name, d in (("FETCH_SIZE", d_f), ("WRITE_SIZE", d_w)):
        acc, n = counters(d); k = pick(acc, "landing_ipm_kernel")
        per[name + "_KB"] = acc[k][name] / n[k]
        per["launches_" + name] = n[k]
    acc, n = counters(d_sq); k = pick(acc, "landing_ipm_kernel")
    out["SQ_per_launch"] = {c: v / n[k] for c, v in acc[k].items()}
    cal = None
    if len(sys.argv) >= 8:
        cal = {}
        for name, d in (("FETCH_SIZE", sys.argv[6]), ("WRITE_SIZE", sys.argv[7])):
            acc, n = counters(d)
            for sub, label, rd, wr in (("AUnaryFunctor<double", "stream8", 1, 1), ("_scatter_gather_elementwise_kernel", "gather8", 2, 1)):
                k = pick(acc, sub)
                if k and (label + "_" + name) not in cal:
                    true_kb = (1 << 27) * 8 / 1024.0 * (rd if name == "FETCH_SIZE" else wr)      # the gather also reads its 8-byte indices
                    cal[label + "_" + name] = {"counter_KB_per_launch": acc[k][name] / n[k], "true_KB": true_kb, "counter_over_true": acc[k][name] / n[k] / true_kb, "kernel": k[:60]}
        out["calibration"] = cal
    out.update(per)
    fr, wr = per["FETCH_SIZE_KB"] * 1024.0, per["WRITE_SIZE_KB"] * 1024.0
    cf = cw = 1.0
    if cal:
        g = cal.get("gather8_FETCH_SIZE") or cal.get("stream8_FETCH_SIZE")
        if g: cf = 1.0 / g["counter_over_true"]
        w = cal.get("stream8_WRITE_SIZE")
        if w: cw = 1.0 / w["counter_over_true"]
    out["traffic_bytes_per_launch"] = fr * cf + wr * cw
    out["traffic_raw_bytes_per_launch"] = fr + wr
    out["note"] = ("FETCH_SIZE/WRITE_SIZE are reported in KB; corrected with the 8-byte-access calibration of tools/pmc_calib.py on the same box "
                   "(factors %.3f read, %.3f write; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide streaming reads 2x, other widths need calibration)" % (cf, cw))
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "%s_pmc_ipm.json" % tag), "w"), indent=1)
    print(json.dumps(out, indent=1))
