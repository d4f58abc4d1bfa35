#!/bin/bash
# SQ counters of the partitioned join's kernels on LUBM-8000's two-key join (profiles/tools/lubm_join_bench.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 540 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/lp_sq1 -- python3 $R/profiles/tools/lubm_join_bench.py 8000 > $R/gpurun_out/lp_sq1.log 2>&1 &&
timeout -k 10 540 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/lp_sq2 -- python3 $R/profiles/tools/lubm_join_bench.py 8000 > $R/gpurun_out/lp_sq2.log 2>&1 || exit 1
cd $R
python3 profiles/summarize.py counters gpurun_out/lp_counters.json ${KSUB:-part_} gpurun_out/lp_sq1 gpurun_out/lp_sq2
rm -rf gpurun_out/lp_sq1 gpurun_out/lp_sq2
