"""The link in both directions at once: pinned -> device and device -> pinned copies of 1 GiB, each alone and together
(two streams).  What bounds a stream front end that sends rows AND decoded qualities back while the next chunks come in."""
import time
import torch
n = 1 << 30
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_out = torch.ones(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(h2d, d2h, frac_back=1.0):
    torch.cuda.synchronize()
    m = int(n * frac_back)
    t0 = time.perf_counter()
    if h2d:
        with torch.cuda.stream(s1):
            d_in.copy_(h_in, non_blocking=True)
    if d2h:
        with torch.cuda.stream(s2):
            h_out[:m].copy_(d_out[:m], non_blocking=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for _ in range(2):
    run(True, True)
for name, a, b, f in (("host -> device alone", True, False, 1.0), ("device -> host alone", False, True, 1.0),
                      ("both, 1 GiB each way", True, True, 1.0), ("both, 0.67 GiB back per GiB in (the decode stream's ratio)", True, True, 0.67),
                      ("both, 0.15 GiB back per GiB in (rows only)", True, True, 0.15)):
    best = min(run(a, b, f) for _ in range(5))
    moved = (n if a else 0) + (int(n * f) if b else 0)
    print("%-62s %6.2f ms  in %5.1f GB/s  total %5.1f GB/s" % (name, best * 1e3, (n / best / 1e9) if a else 0.0, moved / best / 1e9), flush=True)
