set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats -d $O/prof_tr -o tr -- python3 bench.py --workload train --steps 5 --warmup 2 > $O/r3_train_prof2.json 2> $O/r3_train_prof2.err
python tests/prof_summary.py $O/prof_tr/tr_results.db 7 $O/r3_train_kernel_stats2.csv > $O/r3_train_prof_summary2.txt
rm -rf $O/prof_tr
head -42 $O/r3_train_prof_summary2.txt | cut -c1-140
