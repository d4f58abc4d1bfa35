#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/nc_trace -- python3 $R/bench.py --no-scan --no-cpu > $R/gpurun_out/nc_trace.json 2> $R/gpurun_out/nc_trace.err
cd $R
grep "no-table-cache step" gpurun_out/nc_trace.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/nc_trace/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(d.items(), key=lambda kv: -max(kv[1]))[:8]:
    v2 = sorted(v)
    print(k, "n", len(v), "median ms", round(v2[len(v2) // 2], 3), "max ms", round(v2[-1], 3), "top5", [round(x, 2) for x in v2[-5:]])
PY
rm -rf gpurun_out/nc_trace
