#!/bin/bash
# rocprofv3 kernel trace of profiles/tools/lubm_join_bench.py: per-kernel stats + the dispatch sequence of one partitioned join
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp_trace -- python3 $R/profiles/tools/lubm_join_bench.py 8000 > $R/gpurun_out/pp.log 2>&1 || exit 1
cd $R
find gpurun_out/pp_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/pp_kernel_stats.csv \;
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/pp_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
joins = [i for i, r in enumerate(rows) if "part_join_kernel" in r[2]]
out = open("gpurun_out/pp_sequence.txt", "w")
for j in joins[-8:]:                       # the dispatches that lead up to each of the last partitioned joins
    i = j
    while i > 0 and ("part_" in rows[i - 1][2] or "small_scan" in rows[i - 1][2] or "band_bounds" in rows[i - 1][2]): i -= 1
    t0 = rows[i][0]
    out.write(f"--- join ending at dispatch {j}: {(rows[j][1] - t0) / 1e3:.1f} us from its first partition kernel\n")
    for s, e, n in rows[i:j + 1]: out.write(f"  {(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  {n[-60:]}\n")
out.close()
PY
rm -rf gpurun_out/pp_trace
