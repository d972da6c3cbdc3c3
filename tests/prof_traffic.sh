# HBM traffic of the swap step's conv launches (PMC passes only; tests/prof_all.sh is the full capture)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_t_$c -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_t_$c.json 2> $O/pmc_t_$c.err
  echo "pmc $c done"
done
COMMIT=$(cat tests/.commit 2>/dev/null || echo unknown)
python tests/pmc_summary.py $O/pmc_t_ $O/t_pmc_traffic.json "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline" $COMMIT > $O/t_pmc_traffic.txt
rm -rf $O/pmc_t_FETCH_SIZE $O/pmc_t_WRITE_SIZE $O/pmc_t_GRBM_GUI_ACTIVE
head -20 $O/t_pmc_traffic.txt
