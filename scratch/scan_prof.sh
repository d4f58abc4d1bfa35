#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/sp_sq1 -- python3 $R/scratch/scan_prof.py > $R/gpurun_out/sp_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/sp_sq2 -- python3 $R/scratch/scan_prof.py > $R/gpurun_out/sp_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/sp_fetch -- python3 $R/scratch/scan_prof.py > $R/gpurun_out/sp_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/sp_write -- python3 $R/scratch/scan_prof.py > $R/gpurun_out/sp_write.log 2>&1
cd $R
python3 profiles/summarize.py counters gpurun_out/sp_counters.json filter_kernel gpurun_out/sp_sq1 gpurun_out/sp_sq2 gpurun_out/sp_fetch gpurun_out/sp_write
rm -rf gpurun_out/sp_sq1 gpurun_out/sp_sq2 gpurun_out/sp_fetch gpurun_out/sp_write
