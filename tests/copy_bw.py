import torch
x = torch.randn(8,512,512,32, device="cuda"); y = torch.empty_like(x)
for _ in range(5): y.copy_(x)
e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): y.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/50
print("copy 268 MB -> 268 MB: %.1f us = %.0f GB/s" % (ms*1e3, 2*x.numel()*4/ms/1e6))
