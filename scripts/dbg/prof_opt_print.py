"""prof_opt.sh's printer: the top of a rocprofv3 kernel_stats.csv, and the last launches of one kernel from the kernel_trace.csv."""
import csv
import sys

stats, trace, kernel = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
for r in list(csv.DictReader(open(stats)))[:20]:
    print("%-70s calls %6s  avg %10.1f us" % (r["Name"].replace("clstm::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
if kernel:
    rows = [r for r in csv.DictReader(open(trace)) if kernel in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    grid = [r.get("Grid_Size_X", r.get("Grid_Size", "?")) for r in rows]
    print(kernel, "launches", len(d), "last 8 durations (us):", [round(x, 1) for x in d[-8:]], "grid x of the last 4:", grid[-4:])
