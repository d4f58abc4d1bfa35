#!/bin/bash
# The profile set of a round for the default bench command (N = 1): kernel trace + stats of `python3 bench.py`, FETCH_SIZE and
# WRITE_SIZE each in its own PMC pass (no trace domain beside --pmc), and the same two passes for the two scaled scan + FILTER
# workloads; summarised by profiles/summarize.py (which stamps the device sources' hash: bench.py quotes a summary only while it matches).
# usage (on the GPU box): bash profiles/tools/round_profiles.sh <tag>      -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_trace -- python3 $R/bench.py > $R/gpurun_out/${TAG}_bench_n1_under_rocprof.json 2> $R/gpurun_out/${TAG}_trace.log || exit 1
echo "trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pf_fetch -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu --no-cold > $R/gpurun_out/${TAG}_fetch.log 2>&1 || exit 2
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pf_write -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu --no-cold > $R/gpurun_out/${TAG}_write.log 2>&1 || exit 3
echo "write done"
cd $R
python3 profiles/summarize.py stats gpurun_out/pf_trace gpurun_out/${TAG}_bench_kernel_stats.csv || exit 4
python3 profiles/summarize.py pmc gpurun_out/pf_fetch gpurun_out/pf_write gpurun_out/${TAG}_pmc_fetch_write_per_kernel.json queries=262144 products=285000 || exit 5
rm -rf gpurun_out/pf_trace gpurun_out/pf_fetch gpurun_out/pf_write
# the scaled scan + FILTER alone (2000 distinct literals, then a 4 M-literal dictionary): its own PMC passes
for D in 2000 4194304; do
  export DISTINCT=$D
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pf_sfetch -- python3 $R/profiles/tools/scan_prof.py > $R/gpurun_out/${TAG}_scan_fetch_$D.log 2>&1 || exit 6
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pf_swrite -- python3 $R/profiles/tools/scan_prof.py > $R/gpurun_out/${TAG}_scan_write_$D.log 2>&1 || exit 7
  cd $R
  python3 profiles/summarize.py pmc gpurun_out/pf_sfetch gpurun_out/pf_swrite gpurun_out/${TAG}_scan${D}_pmc_fetch_write_per_kernel.json scan_rows=67108864 distinct=$D || exit 8
  rm -rf gpurun_out/pf_sfetch gpurun_out/pf_swrite
  echo "scan $D done"
done
