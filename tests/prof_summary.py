"""Summarise a rocprofv3 results .db (kernel trace): per-kernel totals and per-(kernel, grid) lines.
   python tests/prof_summary.py gpurun_out/prof_x/r01_results.db [steps_in_run] [csv_out]"""
import csv, re, sqlite3, sys
db, nsteps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.3f" % (100.0 * r[2] / tot), r[4], r[5]])
print("total kernel time %.3f ms (%.3f ms / step)" % (tot / 1e6, tot / 1e6 / nsteps))
for r in rows[:28]:
    print("%-60s n/step %6.1f  %8.3f ms/step %6.2f%%  avg %8.1f us" % (re.sub(r"\(.*", "", r[0])[:60], r[1] / nsteps, r[2] / 1e6 / nsteps, 100.0 * r[2] / tot, r[3] / 1e3))
print()
rows = list(c.execute("select name, grid_x, grid_y, grid_z, count(*), sum(end-start), avg(end-start) from kernels where name not like '%conv_mfma%' group by name, grid_x, grid_y, grid_z order by 6 desc limit 30"))
for r in rows:
    print("%-44s grid %9d x%5d x%3d  n/step %5.1f  %7.3f ms/step  avg %8.1f us" % (re.sub(r"\(.*", "", r[0])[:44], r[1], r[2], r[3], r[4] / nsteps, r[5] / 1e6 / nsteps, r[6] / 1e3))
