#!/bin/bash
# the device time line of steady-state Q5 steps (no event brackets): busy time, span, launch gaps (profiles/summarize.py gaps)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export B=${B:-262144} TIMING=0 STEPS=${STEPS:-12}
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gp_trace -- python3 $R/profiles/tools/one_batch.py > $R/gpurun_out/gp_trace.log 2>&1 || exit 1
cd $R
python3 profiles/summarize.py gaps gpurun_out/gp_trace gpurun_out/gp_gaps.json ${FIRST:-oj_probe_kernel}
rm -rf gpurun_out/gp_trace
