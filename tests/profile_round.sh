#!/bin/bash
# Profile passes of one round on the GPU box (run through gpurun from the repo root): rocprofv3 kernel statistics and PMC passes
# (one run per counter, --kernel-trace only beside --pmc, the program itself after `--`: MI355X_MICROARCH.md) of bench.py's
# workloads.  The raw databases stay on the box (gpurun merges at most 64 MiB back): the summaries -- the files that go to
# profiles/ -- are made there by tests/prof_summary.py / pmc_summary.py / pmc_mfma_summary.py.
#   tests/profile_round.sh r05 swap|train|train_bf16|train_b8|hires <commit>
set -e
tag=$1; what=$2; commit=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
case $what in
  swap) args="--steps 8 --warmup 1 --no-cpu-baseline --no-extras"; nsteps=9;;
  train) args="--workload train --steps 6 --warmup 1"; nsteps=7;;
  train_bf16) args="--workload train --precision bf16 --steps 6 --warmup 1"; nsteps=7;;
  train_b8) args="--workload train --train-batch 8 --steps 3 --warmup 1"; nsteps=4;;
  hires) args="--workload hires --precision fp16 --batch 4 --steps 6 --warmup 1"; nsteps=7;;
esac
raw=/tmp/prof_${tag}_${what}
out=gpurun_out/${tag}_bench_${what}
rm -rf $raw; mkdir -p $raw
rocprofv3 --kernel-trace --stats -d $raw/stats -o prof -- python3 bench.py $args > ${out}_under_rocprof.json 2> $raw/stats.err
db=$(ls $raw/stats/*.db | head -1)
python3 tests/prof_summary.py $db $nsteps ${out}_kernel_stats.csv > ${out}_prof_summary.txt
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c -d $raw/pmc_$c -o pmc -- python3 bench.py $args > $raw/pmc_$c.json 2> $raw/pmc_$c.err
  mv $raw/pmc_$c/*.db $raw/pmc_$c/pmc_results.db 2>/dev/null || true
done
python3 tests/pmc_summary.py $raw/pmc_ ${out}_pmc_traffic.json "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE> -- python3 bench.py $args" $commit > ${out}_pmc_traffic.txt
mkdir -p $raw/mf; ln -sfn $raw/pmc_SQ_VALU_MFMA_BUSY_CYCLES $raw/mf_SQ_VALU_MFMA_BUSY_CYCLES; ln -sfn $raw/pmc_GRBM_GUI_ACTIVE $raw/mf_GRBM_GUI_ACTIVE
python3 tests/pmc_mfma_summary.py $raw/mf_ ${out}_pmc_mfma.json $commit > ${out}_pmc_mfma.txt
rm -rf $raw
ls -la ${out}_*
