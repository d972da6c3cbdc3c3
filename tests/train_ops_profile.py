"""Tuning aid (GPU): which torch-level ops (not our HIP launches) the train step issues, by count -- glue to trim."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from torch.profiler import profile, ProfilerActivity
sys.argv = ["bench.py", "--workload", "train", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
with profile(activities=[ProfilerActivity.CPU], record_shapes=False, with_stack=True) as prof:
    bench.main()
ev = prof.key_averages(group_by_stack_n=4)
rows = sorted(ev, key=lambda e: -e.count)
seen = 0
for e in rows:
    if not e.key.startswith("aten::"):
        continue
    st = [s for s in e.stack if "ppst_amd" in s or "bench.py" in s][:2]
    print("%6d  %-28s %s" % (e.count, e.key, " <- ".join(s.strip()[-70:] for s in st)))
    seen += 1
    if seen > 45:
        break
