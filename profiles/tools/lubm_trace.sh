cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp_trace -- python3 $R/profiles/tools/lubm_join_bench.py 8000 > $R/gpurun_out/pp.log 2>&1 || exit 1
cd $R
find gpurun_out/pp_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/pp_kernel_stats.csv \;
rm -rf gpurun_out/pp_trace
grep -E "part_|Name|scan|rocprim" gpurun_out/pp_kernel_stats.csv | cut -c1-200
