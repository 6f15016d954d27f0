#!/bin/bash
# kernel-trace timeline of the bench step, with and without the overlapped weight-gradient GEMM
TAG=${1:-tr}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for OV in 0 1; do
  CLSTM_OVERLAP=$OV timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_ov$OV" -o t -- python "$ROOT/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --profile-steps 0 > "$OUT/trace_ov$OV.log" 2>&1
  tail -1 "$OUT/trace_ov$OV.log" | cut -c1-200
  F=$(find "$OUT/trace_ov$OV" -name "*kernel_trace.csv" | head -1)
  python - "$F" "$OUT/timeline_ov$OV.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete step: from the last k_ingest_pack to the following k_update
idx = [i for i, r in enumerate(rows) if "k_ingest_pack" in r["Kernel_Name"]]
a = idx[-2]; b = idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
out = open(sys.argv[2], "w")
out.write("one training step (times in us from the start of k_ingest_pack): start  end  duration  queue  kernel\n")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.write("%9.2f %9.2f %8.2f  q%-3s %s\n" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:70]))
out.write("step length %.2f us\n" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
out.close()
print(open(sys.argv[2]).read())
PY
  find "$OUT/trace_ov$OV" -name "*.csv" -size +4M -delete
done
