import torch, time
for mb in (160, 1024):
    x = torch.empty(mb * 1000 * 1000 // 8, dtype=torch.int64, device="cuda")
    y = torch.empty_like(x)
    for name, fn in (("fill", lambda: x.fill_(7)), ("zero", lambda: x.zero_()), ("copy", lambda: y.copy_(x))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        print("%4d MB %s: %.1f us -> %.2f TB/s (bytes written)" % (mb, name, ms * 1e3, x.numel() * 8 / ms / 1e9))
