#!/bin/bash
# Round-2 profile set of the default bench command (N = 1): kernel trace + stats of `python3 bench.py`, and FETCH_SIZE /
# WRITE_SIZE in separate PMC passes (no trace domains beside --pmc), summarised by profiles/summarize.py.
# usage (on the GPU box): bash scratch/r02_profiles.sh <tag>      -> gpurun_out/<tag>_*
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_trace -- python3 $R/bench.py > $R/gpurun_out/${TAG}_bench_n1_under_rocprof.json 2> $R/gpurun_out/${TAG}_trace.log
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pf_fetch -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu --no-cold > $R/gpurun_out/${TAG}_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pf_write -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu --no-cold > $R/gpurun_out/${TAG}_write.log 2>&1
echo "write done"
cd $R
python3 profiles/summarize.py stats gpurun_out/pf_trace gpurun_out/${TAG}_bench_kernel_stats.csv
python3 profiles/summarize.py pmc gpurun_out/pf_fetch gpurun_out/pf_write gpurun_out/${TAG}_pmc_fetch_write_per_kernel.json queries=262144 products=285000
# the scaled scan + FILTER alone (2000 distinct literals, then a 4 M-literal dictionary): its own PMC passes
for D in 2000 4194304; do
  export DISTINCT=$D
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pf_sfetch -- python3 $R/scratch/scan_prof.py > $R/gpurun_out/${TAG}_scan_fetch_$D.log 2>&1)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pf_swrite -- python3 $R/scratch/scan_prof.py > $R/gpurun_out/${TAG}_scan_write_$D.log 2>&1)
  python3 profiles/summarize.py pmc gpurun_out/pf_sfetch gpurun_out/pf_swrite gpurun_out/${TAG}_scan${D}_pmc_fetch_write_per_kernel.json scan_rows=67108864 distinct=$D
  rm -rf gpurun_out/pf_sfetch gpurun_out/pf_swrite
done
rm -rf gpurun_out/pf_trace gpurun_out/pf_fetch gpurun_out/pf_write
