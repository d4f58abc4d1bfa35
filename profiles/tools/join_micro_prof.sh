cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export NB=1000
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/jm1 -- python3 $R/profiles/tools/join_micro.py > $R/gpurun_out/jm1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/jm2 -- python3 $R/profiles/tools/join_micro.py > $R/gpurun_out/jm2.log 2>&1 || exit 1
cd $R
python3 profiles/summarize.py counters gpurun_out/jm_counters.json lds_join gpurun_out/jm1 gpurun_out/jm2 > gpurun_out/jm_sum.log 2>&1
rm -rf gpurun_out/jm1 gpurun_out/jm2
