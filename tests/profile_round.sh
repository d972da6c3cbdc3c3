#!/bin/bash
# Profile passes of one round on the GPU box (run through gpurun from the repo root): rocprofv3 kernel statistics and PMC passes
# (one run per counter, --kernel-trace only beside --pmc, the program itself after `--`: MI355X_MICROARCH.md) of bench.py's
# workloads; raw output under gpurun_out/, summaries for profiles/ by tests/pmc_summary.py / pmc_mfma_summary.py / prof_summary.py.
#   tests/profile_round.sh r04 swap|train|train_bf16|hires
set -e
tag=$1; what=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
case $what in
  swap) args="--steps 8 --warmup 1 --no-cpu-baseline --no-extras";;
  train) args="--workload train --steps 6 --warmup 1";;
  train_bf16) args="--workload train --precision bf16 --steps 6 --warmup 1";;
  hires) args="--workload hires --precision fp16 --batch 4 --steps 6 --warmup 1";;
esac
out=gpurun_out/${tag}_${what}
rocprofv3 --kernel-trace --stats -d ${out}_stats -o prof -- python3 bench.py $args > ${out}_under_rocprof.json 2> ${out}_stats.err
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c -d ${out}_pmc_$c -o pmc -- python3 bench.py $args > ${out}_pmc_$c.json 2> ${out}_pmc_$c.err
done
ls ${out}_stats ${out}_pmc_FETCH_SIZE
