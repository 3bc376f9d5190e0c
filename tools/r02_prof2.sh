# Round-2 evidence, part 2: the other BASELINE configs and their rocprof stats.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
{
echo "== config 3: ViT-B/16 384^2 B=128"; python3 $R/bench.py --arch vit_base --size 384 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-slab 2>/dev/null
echo "== config 3 in bf16 mode"; python3 $R/bench.py --arch vit_base --size 384 --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-slab --precision bf16 2>/dev/null
echo "== config 4: slab sweep (default precision)"; python3 $R/tools/sweep_slab.py 2>/dev/null
echo "== config 5: Swin-T"; python3 $R/tools/bench_swin.py 2>/dev/null
echo "== one tile per call"; python3 $R/tools/latency_b1.py bf16x3 2>/dev/null; python3 $R/tools/latency_b1.py bf16 2>/dev/null
} > $O/other_configs.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vitb -- python3 $R/bench.py --arch vit_base --size 384 --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-slab > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slab -- python3 $R/tools/sweep_slab.py --reps 1 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_swin -- python3 $R/tools/bench_swin.py > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_slab -- python3 $R/tools/sweep_slab.py --reps 1 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_slab -- python3 $R/tools/sweep_slab.py --reps 1 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq_swin -- python3 $R/tools/bench_swin.py > /dev/null 2>&1
cd $R
for n in vitb slab swin; do find $O/prof_$n -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$n.csv \; ; done
for d in pmc_fetch_slab pmc_write_slab pmc_sq_swin; do python3 tools/pmc_summary.py $O/$d > $O/$d.txt 2>&1 < /dev/null; done
cat $O/other_configs.txt | cut -c1-400
