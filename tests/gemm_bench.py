"""Tuning aid: the correspondence GEMMs (ppst_model.py:363 / :385) per kernel -- fp32 MFMA against the split-bf16 forms.
python tests/gemm_bench.py   (GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppst_amd import ops


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = "cuda:0"
    torch.manual_seed(0)
    for name, nt, (b, M, N, K) in [("corr  q.k^T", True, (8, 4096, 4096, 512)), ("warp  P.V", False, (8, 4096, 480, 4096)),
                                   ("warp  P.patches", False, (8, 4096, 192, 4096)), ("train P.V", False, (2, 4096, 480, 4096)),
                                   ("train dP = g.V^T", True, (2, 4096, 4096, 64))]:
        A = torch.randn(b, M, K, device=dev)
        Bm = torch.randn(b, N, K, device=dev) if nt else torch.randn(b, K, N, device=dev)
        ref = None
        line = "%-18s b=%d %dx%dx%d:" % (name, b, M, N, K)
        for mode in ("f32", "x6", "x3"):
            f = (lambda: ops.gemm_nt(A, Bm, 1.0, mode=mode)) if nt else (lambda: ops.gemm_nn(A, Bm, mode=mode))
            ms = timeit(f)
            out = f()
            if ref is None:
                ref = out
            err = float((out - ref).abs().max() / ref.abs().max())
            line += "  %s %.3f ms (%.0f TFLOP/s, vs f32 %.1e)" % (mode, ms, 2.0 * b * M * N * K / ms / 1e9, err)
        print(line, flush=True)


if __name__ == "__main__":
    main()
