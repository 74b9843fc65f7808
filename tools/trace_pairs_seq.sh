#!/bin/bash
# The kernels of ONE gf_scan_pairs_device call in dispatch order with their durations (GPU box).
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_pairs_seq
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/tools/bench_pairs.py --profile-mode --steps 2 > /dev/null 2>&1
cd $REPO
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0])))
rows = [r for r in rows if "gf_k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("void ", "").split("(")[0] for r in rows]
last = max(i for i, n in enumerate(names) if n.startswith("gf_k_merge_find"))
t0 = int(rows[last]["Start_Timestamp"])
for i in range(last, len(rows)):
    if names[i].startswith("gf_k_pair_hits_finish") or (i > last and names[i].startswith(("gf_k_fq_", "gf_k_merge_find"))): break
    r = rows[i]
    print("%8.3f ms  %-40s %8.1f us  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, names[i][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
PY
rm -rf $OUT
