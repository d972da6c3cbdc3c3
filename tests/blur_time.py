"""Tuning aid: time of the NHWC blur (ppst_blur_nhwc) on the train step's discriminator shapes, bf16 / fp32 storage.
  python tests/blur_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ppst_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
k4 = torch.tensor([1., 3., 3., 1.]); k4 = (k4[:, None] * k4[None, :]); k4 = (k4 / k4.sum()).to(dev)
k3 = torch.tensor([1., 2., 1.]); k3 = (k3[:, None] * k3[None, :]); k3 = (k3 / k3.sum()).to(dev)
N = 200
print("%-34s %10s %10s" % ("blur", "fp32 us", "bf16 us"))
for name, B, H, C, k, pad, kw in [("4x4 s2d 512^2 x 64 B4", 4, 512, 64, k4, (2, 2), dict(s2d=True)), ("4x4 s2d 256^2 x 128 B4", 4, 256, 128, k4, (2, 2), dict(s2d=True)),
                                  ("4x4 s2d 128^2 x 256 B4", 4, 128, 256, k4, (2, 2), dict(s2d=True)), ("4x4 s2d 64^2 x 512 B4", 4, 64, 512, k4, (2, 2), dict(s2d=True)),
                                  ("4x4 down 2 512^2 x 64 B4", 4, 512, 64, k4, (1, 1), dict(down=2)), ("4x4 plain 513^2 x 64 B2", 2, 513, 64, k4, (2, 2), dict()),
                                  ("3x3 s2d 512^2 x 32 B2", 2, 512, 32, k3, (1, 1), dict(s2d=True))]:
    row = []
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(B, H, H, C, device=dev).to(dt)
        for _ in range(5):
            ops.blur_nhwc(x, k, pad[0], pad[1], ops.PAD_ZERO, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            ops.blur_nhwc(x, k, pad[0], pad[1], ops.PAD_ZERO, **kw)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / N * 1e3)
    print("%-34s %10.1f %10.1f" % (name, *row), flush=True)
