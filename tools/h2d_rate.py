import torch, time
n = 160 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(3): d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): d.copy_(h, non_blocking=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("pinned H2D 160 MiB: %.2f ms = %.1f GB/s" % (ms, n / ms / 1e6))
