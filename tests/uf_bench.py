"""Tuning aid: the blur of the encoders' 512^2 x 32 maps (bench.py's roofline_upfirdn2d) alone, several times.
PPST_HIP_LIB=<variant .so> python tests/uf_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

if __name__ == "__main__":
    for B in (8, 24):
        r = [bench.upfirdn2d_rate(B, torch.device("cuda", 0)) for _ in range(5)]
        print("B=%d" % B, " ".join("%.0f" % x["achieved"] for x in r), "GB/s;", r[-1]["us_per_launch"], "us", flush=True)
