# SQ counters of the split-bf16 GEMM variants (two --pmc passes over tools/microbench_x3.py). Usage on the GPU box:
#   bash tools/pmc_x3.sh "<microbench args>" <tag>
R=$GRAFT_REPO_ROOT
ARGS="$1"; TAG="$2"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq1 -- python3 $R/tools/microbench_x3.py $ARGS > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq2 -- python3 $R/tools/microbench_x3.py $ARGS > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq3 -- python3 $R/tools/microbench_x3.py $ARGS > /dev/null 2>&1
cd $R
for i in 1 2 3; do python3 tools/pmc_summary.py gpurun_out/${TAG}_sq$i > gpurun_out/${TAG}_sq$i.txt 2>&1 < /dev/null; done
cat gpurun_out/${TAG}_sq1.txt gpurun_out/${TAG}_sq2.txt gpurun_out/${TAG}_sq3.txt | grep -v "^$" | cut -c1-260
