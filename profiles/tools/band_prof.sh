#!/bin/bash
# rocprofv3 passes over one batched BSBM-100M Q5 workload (profiles/tools/one_batch.py, B instances): kernel trace + SQ counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${B:-262144}
export B
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bp_trace -- python3 $R/profiles/tools/one_batch.py > $R/gpurun_out/bp_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/bp_sq1 -- python3 $R/profiles/tools/one_batch.py > $R/gpurun_out/bp_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/bp_sq2 -- python3 $R/profiles/tools/one_batch.py > $R/gpurun_out/bp_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/bp_fetch -- python3 $R/profiles/tools/one_batch.py > $R/gpurun_out/bp_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/bp_write -- python3 $R/profiles/tools/one_batch.py > $R/gpurun_out/bp_write.log 2>&1
cd $R
python3 profiles/summarize.py counters gpurun_out/bp_counters.json ${KSUB:-band_} gpurun_out/bp_sq1 gpurun_out/bp_sq2 gpurun_out/bp_fetch gpurun_out/bp_write
find gpurun_out/bp_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/bp_kernel_stats.csv \;
rm -rf gpurun_out/bp_trace gpurun_out/bp_sq1 gpurun_out/bp_sq2 gpurun_out/bp_fetch gpurun_out/bp_write
