set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python bench.py --steps 20 --warmup 3 > $O/r03_bench.json 2> $O/r03_bench.err
echo "bench done" ; cut -c1-300 $O/r03_bench.json
rocprofv3 --kernel-trace --stats -d $O/prof_r03 -o sw -- python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline > $O/r03_bench_under_rocprof.json 2> $O/r03_rocprof.err
python tests/prof_summary.py $O/prof_r03/sw_results.db 9 $O/r03_bench_kernel_stats.csv > $O/r03_bench_prof_summary.txt
rm -rf $O/prof_r03
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_r03_$c -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_r03_$c.json 2> $O/pmc_r03_$c.err
  echo "pmc $c done"
done
COMMIT=$(cat tests/.commit 2>/dev/null || echo unknown)
python tests/pmc_summary.py $O/pmc_r03_ $O/r03_pmc_traffic.json "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline" $COMMIT > $O/r03_pmc_traffic.txt
python tests/pmc_mfma_summary.py $O/pmc_r03_ $O/r03_pmc_mfma.json $COMMIT > $O/r03_pmc_mfma.txt
rm -rf $O/pmc_r03_FETCH_SIZE $O/pmc_r03_WRITE_SIZE $O/pmc_r03_GRBM_GUI_ACTIVE $O/pmc_r03_SQ_VALU_MFMA_BUSY_CYCLES
python bench.py --workload train --steps 5 --warmup 2 > $O/r03_bench_train.json 2> $O/r03_bench_train.err
rocprofv3 --kernel-trace --stats -d $O/prof_r03_train -o tr -- python3 bench.py --workload train --steps 5 --warmup 2 > $O/r03_bench_train_under_rocprof.json 2> $O/r03_rocprof_train.err
python tests/prof_summary.py $O/prof_r03_train/tr_results.db 7 $O/r03_train_kernel_stats.csv > $O/r03_train_prof_summary.txt
rm -rf $O/prof_r03_train
python bench.py --workload train --precision bf16 --steps 5 --warmup 2 > $O/r03_bench_train_bf16.json 2> $O/r03_bench_train_bf16.err
python bench.py --workload hires --precision fp16 --batch 4 --steps 5 --warmup 2 > $O/r03_bench_hires_fp16.json 2> $O/r03_bench_hires.err
python bench.py --workload hires --batch 4 --steps 5 --warmup 2 > $O/r03_bench_hires_bf16x3.json 2>> $O/r03_bench_hires.err
python bench.py --workload grid --steps 3 --warmup 1 > $O/r03_bench_grid.json 2> $O/r03_bench_grid.err
python bench.py --precision fp16 --steps 10 --warmup 2 --no-cpu-baseline > $O/r03_bench_fp16.json 2> /dev/null
echo ALL DONE
