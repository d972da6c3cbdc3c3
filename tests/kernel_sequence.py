"""Tuning aid: the ORDERED kernel launches of the last step of a rocprofv3 kernel trace (results .db), one line per launch with its
grid, duration and the gap to the previous kernel -- to see which producer feeds which consumer (round 5: the swap step's tail).
   python tests/kernel_sequence.py <results.db> <steps_in_run (incl. warm-up)> [min_us]"""
import re, sqlite3, sys
db, nsteps = sys.argv[1], int(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
c = sqlite3.connect(db)
rows = list(c.execute("select name, grid_x, grid_y, grid_z, start, end from kernels order by start"))
per = len(rows) // nsteps
last = rows[len(rows) - per:]
print("%d launches in the run, %d per step; last step: %.3f ms of kernels, %.3f ms first start to last end" % (
    len(rows), per, sum(r[5] - r[4] for r in last) / 1e6, (last[-1][5] - last[0][4]) / 1e6))
prev = None
for i, (name, gx, gy, gz, st, en) in enumerate(last):
    us = (en - st) / 1e3
    if us >= min_us:
        print("%4d %-64s grid %9d x%4d x%3d %9.1f us  gap %6.1f" % (i, re.sub(r"\(.*", "", name)[:64], gx, gy, gz, us, 0.0 if prev is None else (st - prev) / 1e3))
    prev = en
