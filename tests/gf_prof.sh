cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 100 python tests/gf_time.py 4 1024
timeout -k 10 200 python tests/gpu_diag.py guided 2>&1 | tail -8
rm -rf /tmp/gfprof; rocprofv3 --kernel-trace --stats -d /tmp/gfprof -o gf -- python3 tests/gf_time.py 4 1024 > /dev/null 2>&1
python3 - <<'PY'
import sqlite3, glob
db = glob.glob("/tmp/gfprof/*.db")[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
q = "select s.kernel_name, count(*), avg(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (kt, ks)
for n, cnt, av in c.execute(q):
    if "gf_" in n: print("%-60s n %3d avg %.1f us" % (n[:60], cnt, av / 1e3))
PY
