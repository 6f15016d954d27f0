"""one training step out of a rocprofv3 --kernel-trace csv: start, end, duration of every launch (usage: step_timeline.py trace.csv out.txt)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step ends with its last slab reduction (k_reduce_scatter / k_reduce_scatter_ingest: the update, and -- in a loop that declares its
# next minibatch, clstm_net_train_step_next -- the next step's ingest): the launch behind it is the first of the next step
red = lambda r: "clstm::k_reduce_scatter" in r["Kernel_Name"]
idx = [i for i in range(1, len(rows)) if red(rows[i - 1]) and not red(rows[i])]
# the step of MEDIAN length (the first step behind every fence of the timed blocks and the host-paced enqueue burst at the end are longer)
length = lambda j: int(rows[idx[j + 1]]["Start_Timestamp"]) - int(rows[idx[j]]["Start_Timestamp"])
j = sorted(range(len(idx) - 1), key=length)[(len(idx) - 1) // 2]
a, b = idx[j], idx[j + 1]
t0 = int(rows[a]["Start_Timestamp"])
out = open(sys.argv[2], "w")
out.write("one training step under rocprofv3 --kernel-trace (us from the start of the step's first launch): start  end  duration  kernel\n")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.write("%9.2f %9.2f %8.2f  %s\n" % (s / 1e3, e / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:80]))
out.write("step length %.2f us\n" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
out.close()
print(open(sys.argv[2]).read())
