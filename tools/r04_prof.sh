# Round-4 evidence set (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats, PMC passes, other configs.
#   bash tools/r04_prof.sh
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
python3 $R/bench.py --precision bf16 --no-cpu-baseline --no-slab > $O/bench_bf16.json 2>> $O/bench.err
python3 $R/bench.py --precision fp32 --no-cpu-baseline --no-slab --steps 10 > $O/bench_fp32.json 2>> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-slab > $O/prof_bench.json 2> $O/prof.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-slab > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-slab > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-slab > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-slab > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-slab > /dev/null 2>&1
{
echo "== config 3: ViT-B/16 384^2 B=128"; python3 $R/bench.py --arch vit_base --size 384 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-slab 2>/dev/null
echo "== config 4: slab sweep (default precision)"; python3 $R/tools/sweep_slab.py 2>/dev/null
echo "== config 5: Swin-T (bf16 / split-bf16 / fp32)"; for p in bf16 bf16x3 fp32; do python3 $R/tools/bench_swin.py --precision $p 2>/dev/null; done
echo "== one tile per call"; python3 $R/tools/latency_b1.py bf16x3 2>/dev/null
} > $O/other_configs.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vitb -- python3 $R/bench.py --arch vit_base --size 384 --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-slab > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slab -- python3 $R/tools/sweep_slab.py --reps 1 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_slab -- python3 $R/tools/sweep_slab.py --reps 1 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_slab -- python3 $R/tools/sweep_slab.py --reps 1 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_swin -- python3 $R/tools/bench_swin.py --precision bf16x3 --steps 5 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python3 $R/tools/b1_loop.py vit_small 16 224 100 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1_s8 -- python3 $R/tools/b1_loop.py vit_small 8 384 50 > /dev/null 2>&1
cd $R
for d in pmc_fetch pmc_write pmc_sq1 pmc_sq2 pmc_tcc pmc_fetch_slab pmc_write_slab; do python3 tools/pmc_summary.py $O/$d > $O/$d.txt 2>&1 < /dev/null; done
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
for n in vitb slab swin b1 b1_s8; do find $O/prof_$n -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$n.csv \; ; done
rm -rf $O/prof $O/prof_vitb $O/prof_slab $O/prof_swin $O/prof_b1 $O/prof_b1_s8 $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2 $O/pmc_tcc $O/pmc_fetch_slab $O/pmc_write_slab
ls -la $O | head -40
tail -c 400 $O/bench.json < /dev/null
