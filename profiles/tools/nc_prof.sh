#!/bin/bash
# SQ counters of the kernels of a Q5 step with NO cached join table (profiles/tools/nc_kernels.py): the partitioned join that
# writes the 0.54 G-row candidate table and the two generic joins that filter it.   KSUB=<kernel name fragment> (default: _join_kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/nc_sq1 -- python3 $R/profiles/tools/nc_kernels.py > $R/gpurun_out/nc_sq1.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/nc_sq2 -- python3 $R/profiles/tools/nc_kernels.py > $R/gpurun_out/nc_sq2.log 2>&1 || exit 1
cd $R
python3 profiles/summarize.py counters gpurun_out/nc_counters.json ${KSUB:-_join_kernel} gpurun_out/nc_sq1 gpurun_out/nc_sq2
rm -rf gpurun_out/nc_sq1 gpurun_out/nc_sq2
