#!/bin/bash
# kernel timeline of the last steady-state batched Q5 execution (gaps between consecutive GPU operations)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${B:-262144}; export B
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/tl -- python3 $R/scratch/one_batch.py > $R/gpurun_out/tl.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(f))), key=lambda x: x[0])
# the last execution: from the last oj_probe_kernel on
idx = max(i for i, r in enumerate(rows) if "oj_probe" in r[2])
start = idx
while start > 0 and rows[start][0] - rows[start - 1][1] < 200_000: start -= 1
seq = rows[start:]
t0 = seq[0][0]
prev_end = None
busy = 0
for s, e, n in seq:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:7.1f} us  {n}")
    prev_end = e; busy += e - s
print("span", (seq[-1][1] - t0) / 1e3, "us; busy", busy / 1e3, "us; ops", len(seq))
PY
rm -rf gpurun_out/tl
