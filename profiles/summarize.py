"""Turns rocprofv3 output directories into the small per-kernel summaries committed next to this file.

  python profiles/summarize.py stats <dir with *_kernel_stats.csv> <out.csv>
  python profiles/summarize.py pmc   <fetch dir> <write dir> <out.json> [key=value ...]   (key=value -> "_workload", what bench.py matches on;
                                                                                            "_source_sha16" = the device sources' hash, stamped here)
  python profiles/summarize.py counters <out.json> <kernel substring> <dir> [<dir> ...]
  python profiles/summarize.py gaps  <dir with *_kernel_trace.csv> <out.json> <substring of a step's first kernel>   (busy time, span and
                                                                                            launch gaps of the steady-state steps)

The pmc form reads *_counter_collection.csv of two separate passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950) and writes, per kernel: dispatches, median/max raw counter values in KB, and `hbm_bytes_per_launch`
= 2 x FETCH + WRITE  for kernels whose reads are wide streaming loads (the gfx950 FETCH_SIZE halving of
MI355X_MICROARCH.md, HBM section) and FETCH + WRITE otherwise — the `fetch_factor` used is recorded.
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys
from collections import defaultdict

# coalesced streaming readers: 16 B per lane (the guide's case), and run_copy_kernel (4 B per lane, four loads in flight),
# calibrated on its known byte count: 33.5 M survivors x 4 B = 134.1 MB read, FETCH_SIZE reports 67.6 MB
STREAMING = ("filter_kernel", "filter_bits_kernel", "filter_write_kernel", "scan_count_kernel", "scan_write_kernel", "run_copy_kernel")


def short(name):
    return name.split("(")[0].strip()


def source_sha16():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import rdf_fusion_amd
    return rdf_fusion_amd.kernel_source_sha16()


def pmc_values(d, counter):
    out = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out


def main():
    if sys.argv[1] == "stats":
        src = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_stats.csv"), recursive=True)
        shutil.copy(src[0], sys.argv[3])
        return
    if sys.argv[1] == "gaps":       # summarize.py gaps <trace dir> <out.json> <first kernel substring> : device time line of the steady-state steps
        src = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)
        rows = []
        for f in src:
            for r in csv.DictReader(open(f)):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
        rows.sort()
        first = sys.argv[4]
        starts = [i for i, r in enumerate(rows) if first in r[2]]
        steps = []
        for a, b in zip(starts, starts[1:]):           # a step = from one launch of the first kernel to the next ...
            ks = rows[a:b]
            for i in range(len(ks) - 1):               # ... or to the host's pause before it (the driver script prepares the next batch: > 1 ms)
                if ks[i + 1][0] - ks[i][1] > 1_000_000: ks = ks[:i + 1]; b = a + i + 1; break
            busy = sum(e - s for s, e, _ in ks)
            span = ks[-1][1] - ks[0][0]
            gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
            steps.append({"kernels": len(ks), "busy_us": busy / 1e3, "span_us": span / 1e3, "gaps_us": [g / 1e3 for g in gaps],
                          "to_next_step_us": (rows[b][0] - ks[-1][1]) / 1e3, "names": [short(n) for _, _, n in ks]})
        steady = steps[len(steps) // 2:]
        med = lambda v: sorted(v)[len(v) // 2] if v else None
        res = {"steps_seen": len(steps), "steady_steps": len(steady),
               "median_busy_us": med([s["busy_us"] for s in steady]), "median_span_us": med([s["span_us"] for s in steady]),
               "median_gap_sum_us": med([sum(s["gaps_us"]) for s in steady]), "median_between_steps_us": med([s["to_next_step_us"] for s in steady]),
               "last_step": steady[-1] if steady else None, "_source_sha16": source_sha16()}
        json.dump(res, open(sys.argv[3], "w"), indent=1)
        print(json.dumps({k: v for k, v in res.items() if k != "last_step"}))
        return
    if sys.argv[1] == "counters":   # every counter found in the given pass directories, median per dispatch, for kernels matching the substring
        out, want = {}, sys.argv[3]
        for d in sys.argv[4:]:
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                per = defaultdict(lambda: defaultdict(list))
                for r in csv.DictReader(open(f)):
                    if want in r["Kernel_Name"]:
                        per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                for k, cs in per.items():
                    for c, v in cs.items():
                        out.setdefault(k, {})[c] = {"dispatches": len(v), "median": statistics.median(v)}
        json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
        return
    fetch, write = pmc_values(sys.argv[2], "FETCH_SIZE"), pmc_values(sys.argv[3], "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
        factor = 2 if any(s in k for s in STREAMING) else 1
        res[k] = {"dispatches": len(f), "FETCH_SIZE_KB_median": statistics.median(f), "FETCH_SIZE_KB_max": max(f),
                  "WRITE_SIZE_KB_median": statistics.median(w), "WRITE_SIZE_KB_max": max(w), "fetch_factor": factor,
                  "hbm_bytes_per_launch": int(1024 * (factor * statistics.median(f) + statistics.median(w)))}
    if len(sys.argv) > 5:
        res["_workload"] = {kv.split("=", 1)[0]: (int(kv.split("=", 1)[1]) if kv.split("=", 1)[1].lstrip("-").isdigit() else kv.split("=", 1)[1]) for kv in sys.argv[5:]}
    # which kernel code these counters were collected on: bench.py quotes a summary only while the device sources still hash to this
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import rdf_fusion_amd
    res["_source_sha16"] = rdf_fusion_amd.kernel_source_sha16()
    json.dump(res, open(sys.argv[4], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
