R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq1 -- python3 $R/tools/microbench.py --iters 2 --only qkv,proj,fc1,fc2,attention > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq2 -- python3 $R/tools/microbench.py --iters 2 --only qkv,proj,fc1,fc2,attention > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_sq1 > gpurun_out/pmc_sq1.txt 2>&1 < /dev/null
python3 tools/pmc_summary.py gpurun_out/pmc_sq2 > gpurun_out/pmc_sq2.txt 2>&1 < /dev/null
wc -l gpurun_out/pmc_sq1.txt gpurun_out/pmc_sq2.txt
