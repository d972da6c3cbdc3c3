"""Tuning aid: rocprofv3 --kernel-trace CSV -> launches grouped by (kernel, grid, block): count, mean / total duration.
  python tests/kernel_grid_table.py <kernel_trace.csv> <steps> [name filter]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
g = defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    if flt and flt not in n:
        continue
    key = (n[:60], r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Workgroup_Size_X"))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g[key][0] += 1
    g[key][1] += d
tot = sum(v[1] for v in g.values())
print("total %.3f ms / step over %d groups" % (tot / 1e3 / steps, len(g)))
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-60s grid %8s x %2s wg %4s  n/step %6.1f  avg %8.1f us  %7.3f ms/step" % (*k, v[0] / steps, v[1] / v[0], v[1] / 1e3 / steps))
