"""GPU box: write-only / copy bandwidth of torch fills on tensors of the conv layers' sizes (is 157 MB of output in 26 us plausible?)."""
import torch
dev = torch.device("cuda:0")
for mb in (33.5, 157, 629, 2500):
    n = int(mb * 1e6 / 2)
    x = torch.randn(n, device=dev).bfloat16()
    y = torch.empty_like(x)
    for name, fn, byt in (("zero_", lambda: y.zero_(), 2 * n), ("copy_", lambda: y.copy_(x), 4 * n), ("add_", lambda: y.add_(1), 4 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{mb:7.1f} MB {name:6s} {ms * 1e3:8.1f} us  {byt / ms / 1e9:6.2f} TB/s")
