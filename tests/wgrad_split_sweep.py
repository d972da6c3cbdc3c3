"""Tuning aid (GPU): train-step time against the weight gradient's split heuristic (ops.WGRAD_SPLIT)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for blocks, mt in ((1536, 2), (1024, 4), (768, 4), (512, 4), (768, 8), (512, 8), (384, 8)):
    code = ("import sys; sys.argv=['bench.py','--workload','train','--steps','5','--warmup','2','--no-cpu-baseline'];"
            "sys.path.insert(0,%r); from ppst_amd import ops; ops.WGRAD_SPLIT.update(blocks=%d,min_tiles=%d); import bench; bench.main()" % (ROOT, blocks, mt))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT).stdout.strip().splitlines()
    d = json.loads(out[-1])
    print("blocks %5d min_tiles %d : %.1f ms" % (blocks, mt, d["ms_per_step"]), flush=True)
