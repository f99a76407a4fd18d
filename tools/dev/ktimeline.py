"""dev: durations of a kernel's launches in launch order (every STEP-th), from a rocprofv3 --kernel-trace run.   python3 tools/dev/ktimeline.py DIR name [step]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]: rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort(); step = int(sys.argv[3]) if len(sys.argv) > 3 else 20
print(len(rows), "launches; us:", [round(d) for _, d in rows[::step]])
gaps = [(rows[i + 1][0] - rows[i][0]) / 1e3 for i in range(len(rows) - 1)]
print("period between launches, us:", [round(g) for g in gaps[::step]])
