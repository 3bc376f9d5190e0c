set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/final_bench.json 2> $R/gpurun_out/final_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/final_prof_bench.json 2> $R/gpurun_out/final_prof.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final_pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final_pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/final_pmc_fetch > gpurun_out/final_pmc_fetch.txt 2>&1 < /dev/null
python3 tools/pmc_summary.py gpurun_out/final_pmc_write > gpurun_out/final_pmc_write.txt 2>&1 < /dev/null
tail -c 600 gpurun_out/final_bench.json < /dev/null
