"""Context measurement (GPU): the vendor library's own dense-GEMM rate on this box (hipBLASLt through torch.matmul), the
practical ceiling next to which conv_mfma's MFMA fraction should be read.  bf16 and fp32, 8192^3, and a conv-like shape."""
import torch, time
def rate(M, N, K, dt, n=20):
    a = torch.randn(M, K, device="cuda", dtype=dt); b = torch.randn(K, N, device="cuda", dtype=dt)
    for _ in range(3): (a @ b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): (a @ b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return 2.0 * M * N * K / ms / 1e9
for dt in (torch.bfloat16, torch.float16, torch.float32):
    for shp in ((8192, 8192, 8192), (524288, 256, 2304), (131072, 512, 4608)):
        print(dt, shp, "%.1f TFLOP/s" % rate(*shp, dt), flush=True)
